/*
 * daac_synth.h — synthetic haystack generators (bench/test support, not part of the crate's API).
 *
 * Haystacks of 4-32 GiB are generated directly in HBM, as a pure function of (seed, byte index),
 * so nothing crosses PCIe and any shard of a larger haystack can be produced independently
 * (SURVEY.md §8d).  tests/synth_ref.py holds the numpy definition the kernels are checked against.
 *
 *   z(j)  = mix64(seed + (j + 1) * 0x9E3779B97F4A7C15)          (SplitMix64, by index)
 *   uniform:  byte i = alphabet[(b * alphabet_len) >> 8],  b = (z(i / 8) >> (8 * (i % 8))) & 0xff
 *   wordsoup: the haystack is a sequence of `slot_bytes`-byte slots; slot s holds word
 *             w = (z(s) >> 8) % n_words left-aligned and padded with `pad`, unless
 *             (z(s) & 0xff) < noise_256, in which case it holds `slot_bytes - 1` letters
 *             alphabet[...] drawn from z'(s, t) = mix64(z(s) + t) followed by one `pad`.
 *   zipf text (cfg5): `slot_bytes`-byte slots of whole UTF-8 characters; character k of slot s is drawn from
 *             zz = mix64(z(s) + k + 1): an ASCII byte ascii_lo + (((zz >> 8) & 0xffffff) * ascii_n >> 24) when
 *             (zz & 0xff) < ascii_256 or fewer than 3 bytes are left in the slot, otherwise the 3-byte code point
 *             codepoints[i], i = first index with cum_weights[i] > ((zz >> 32) * cum_weights[n - 1]) >> 32.
 */
#ifndef DAAC_SYNTH_H
#define DAAC_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#include "daachorse_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Fills dev_out[0 .. len) with bytes index_offset .. index_offset + len of the uniform stream. */
daac_status daac_synth_uniform(uint8_t *dev_out, size_t len, uint64_t seed, const uint8_t *alphabet,
                               uint32_t alphabet_len, uint64_t index_offset, void *stream);

/* Fills dev_out[0 .. len) with bytes index_offset .. of the word-soup stream.  `words` is a host
 * blob with n_words + 1 host offsets (each word must be shorter than slot_bytes). */
daac_status daac_synth_wordsoup(uint8_t *dev_out, size_t len, uint64_t seed, const uint8_t *words,
                                const uint64_t *offsets, uint32_t n_words, uint32_t slot_bytes, uint8_t pad,
                                uint32_t noise_256, const uint8_t *alphabet, uint32_t alphabet_len,
                                uint64_t index_offset, void *stream);

/* Fills dev_out[0 .. len) with bytes index_offset .. of the Zipf text stream (mixed 1- and 3-byte characters;
 * every multiple of slot_bytes is a character boundary).  `codepoints` / `cum_weights` are host arrays. */
daac_status daac_synth_zipf_text(uint8_t *dev_out, size_t len, uint64_t seed, const uint32_t *codepoints,
                                 const uint32_t *cum_weights, uint32_t n_symbols, uint32_t ascii_256,
                                 uint32_t ascii_lo, uint32_t ascii_n, uint32_t slot_bytes, uint64_t index_offset,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif
