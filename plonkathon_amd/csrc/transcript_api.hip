// transcript_api.hip — host C-ABI for the Fiat-Shamir transcript (plonk_transcript_*).
// Replaces merlin.MerlinTranscript + transcript.py:58-75 on the host; see transcript.h.
#include <string.h>

#include "plonk_internal.h"
#include "transcript.h"

struct plonk_transcript {
    MerlinState s;
};

extern "C" {

int plonk_transcript_new(const uint8_t* label, size_t label_len, plonk_transcript** out) {
    PLONK_REQUIRE(out && (label || !label_len), PLONK_ERR_ARG, "bad argument");
    plonk_transcript* t = new plonk_transcript();
    merlin_init(t->s, label, label_len);
    *out = t;
    return PLONK_OK;
}

int plonk_transcript_clone(const plonk_transcript* t, plonk_transcript** out) {
    PLONK_REQUIRE(t && out, PLONK_ERR_ARG, "bad argument");
    *out = new plonk_transcript(*t);
    return PLONK_OK;
}

int plonk_transcript_free(plonk_transcript* t) {
    delete t;
    return PLONK_OK;
}

int plonk_transcript_append_message(plonk_transcript* t, const uint8_t* label, size_t label_len,
                                    const uint8_t* msg, size_t msg_len) {
    PLONK_REQUIRE(t && (label || !label_len) && (msg || !msg_len), PLONK_ERR_ARG, "bad argument");
    merlin_append_message(t->s, label, label_len, msg, msg_len);
    return PLONK_OK;
}

int plonk_transcript_challenge_bytes(plonk_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out,
                                     size_t n) {
    PLONK_REQUIRE(t && (label || !label_len) && (out || !n), PLONK_ERR_ARG, "bad argument");
    merlin_challenge_bytes(t->s, label, label_len, out, n);
    return PLONK_OK;
}

int plonk_transcript_challenge_scalar(plonk_transcript* t, const uint8_t* label, size_t label_len,
                                      uint8_t out_le32[32]) {
    PLONK_REQUIRE(t && (label || !label_len) && out_le32, PLONK_ERR_ARG, "bad argument");
    Fr f = fp_from_mont(plonk_get_and_append_challenge(t->s, label, label_len));
    memcpy(out_le32, f.v, 32);
    return PLONK_OK;
}

}  // extern "C"
