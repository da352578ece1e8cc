// C shim over the REFERENCE's tokenizer.h (compiled from /root/reference via -I; not copied).
// TEST INFRASTRUCTURE ONLY: used by tests/golden/make_goldens.py to capture encode/decode goldens.
#include <stdio.h>
#include <stdint.h>
#include <ctype.h>
#include "tokenizer.h"   // resolves to $(REF)/tokenizer.h

extern "C" {
void* ref_tok_build(const char* path, int vocab_size) {
    Tokenizer* t = (Tokenizer*)calloc(1, sizeof(Tokenizer));
    build_tokenizer(t, (char*)path, vocab_size);
    return t;
}
void ref_tok_free(void* t) { free_tokenizer((Tokenizer*)t); free(t); }
int ref_tok_encode(void* t, const char* text, int bos, int eos, int* tokens) {
    int n = 0;
    encode((Tokenizer*)t, (char*)text, (int8_t)bos, (int8_t)eos, tokens, &n);
    return n;
}
const char* ref_tok_decode(void* t, int prev, int tok) { return decode((Tokenizer*)t, prev, tok); }
unsigned ref_tok_max_token_length(void* t) { return ((Tokenizer*)t)->max_token_length; }
}
