// llama2_q4 executable: the reference's main() (llama2_q4.cu:622-720) lives behind q4_main in libllama2_q4.so.
#include "llama2_q4.h"
int main(int argc, char** argv) { return q4_main(argc, argv); }
