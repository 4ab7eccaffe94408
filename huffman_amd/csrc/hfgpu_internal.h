// Shared between the host-side codebook code and the HIP translation unit.
#ifndef HFGPU_INTERNAL_H
#define HFGPU_INTERNAL_H

#include <stdint.h>
#include <vector>

// Decode lookup-table geometry.
#define HF_LUT_W1 11u          // index bits of the first level
#define HF_LUT_W2 7u           // index bits of every further level
#define HF_LUT_MAX_SUB 32u     // sub-tables kept in LDS at most
// A 16-bit entry is `symbol << 8 | length` (length <= 57 in the low six bits, so
// the entry itself serves as the shift amount of a 64-bit shift, which reads
// only those bits, and the low byte of a sum of a few entries is the sum of
// their lengths).  Bit 7 marks an escape: `sub-table number << 8 | 0x80`, or
// one of the two special values below.
#define HF_LUT_ESC 0x0080u
#define HF_LUT_ENTRY(sym, len) ((uint32_t)(sym) << 8 | (uint32_t)(len))
#define HF_LUT_LEN(e) ((e) & 0x7fu)
#define HF_LUT_SYM(e) ((e) >> 8)
#define HF_LUT_SEARCH 0xFFFFu  // entry: resolve by binary search over the keys
#define HF_LUT_INVALID 0xFFFEu // entry: window below the smallest key
#define HF_MAX_SENTINEL 57u    // the slice decoder of the reference needs <= 57
#define HF_MAX_CODELEN 57u     // longest code the encoder packs faithfully

namespace hf {

struct DecodeLut {
    uint32_t w1 = 0;
    uint32_t sentinel = 0;
    uint32_t n_sub = 0;
    uint32_t max_direct_len = 0;
    bool needs_search = false;
    bool has_escape = false;    // some entry has the escape bit (sub-table, search, invalid)
    std::vector<uint16_t> pool; // level 1 followed by the sub-tables
};

int code_lengths_sorted(uint64_t *w, uint64_t n);
int sort_nonzero_desc(const uint64_t *store, uint64_t n, uint64_t *idx, uint64_t *cnt, uint64_t *m);
int canonical_codes(const uint64_t *len, uint64_t n, uint64_t *small_codes, uint64_t *big_codes);
int build_codebook(const uint64_t *hist, uint64_t n, uint64_t codes[256], uint64_t lens[256]);
uint64_t sentinel_of(const uint64_t lens[256]);
int model_table(const uint64_t codes[256], const uint64_t lens[256], uint64_t keys[256],
                uint8_t syms[256], uint8_t klens[256], uint64_t *m);
// How many consecutive codes (4, 2 or 1) the encoder merges before one append
// to its bit sink.  A merged group must fit 32 bits; a lane that meets a group
// that does not has to redo its 32 symbols one code at a time through global
// memory, and its wave waits for it.  Returns the widest width for which, with
// symbols drawn with the probabilities 2^-len that the lengths imply, fewer
// than one wave in a thousand has such a lane; `speculative` says whether a
// group can overflow at all.
int pick_merge_width(const uint64_t lens[256], bool *speculative);
// Codes the decoder may take from one 64-bit window without checking (16, 8, 7
// or 6), or 0 when only sentinel-many bits per code are safe.  Only for models
// that are complete prefix codes (a code that fits the window then decodes the
// same whatever follows it, and one that does not fit never decodes as one
// that does): the largest width for which, with symbols drawn with the
// probabilities 2^-len that the lengths imply, fewer than one chunk of 64
// symbols in ten million (16, 8) or in a hundred thousand (7, 6) holds a group
// that exceeds 64 bits.
uint32_t pick_decode_window(const uint64_t *keys, const uint8_t *klens, uint64_t m,
                            uint32_t sentinel);
int build_decode_lut(const uint64_t *keys, const uint8_t *syms, const uint8_t *klens, uint64_t m,
                     uint64_t sentinel, DecodeLut *lut);
// True when the table is a complete prefix code: no code is a prefix of another
// one and every bit window starts with exactly one of them (Kraft sum 1; what
// the Huffman construction yields for two or more symbols).  Only then does a
// window that holds the first v bits of a longer code, whatever follows them,
// never decode as a code of at most v bits: in a table with unused code space
// the predecessor rule maps such a window to the nearest smaller key, which
// can be short.  The decoder's shortcuts that look at windows whose tail is
// not valid (speculative window widths, the two-symbol table) rely on this.
bool is_complete_prefix_code(const uint64_t *keys, const uint8_t *klens, uint64_t m,
                             uint32_t sentinel);
// Two-symbol decode table over the first-level index of `lut` (complete prefix
// codes only).  Entry of index x: sym0 << 8 | sym1 << 16 | (len0 + len1) when
// the two codes that start at the top of x both end inside its w1 bits.
// Otherwise HF_PAIR_SLOW is set: with sym0 << 8 | len0 when only the first
// code ends inside the index (the decoder looks the second one up on its own),
// with HF_PAIR_UNRESOLVED when the first level cannot resolve the first code
// either.  The table is only worth having when few indices are slow: under a
// well-compressed stream every index is about equally likely, so the slow
// fraction of the index space is the probability that a lookup needs the
// fix-up.  Returns false, with `pair` empty, when it exceeds `max_slow`.
// Behind the 1 << w1 first-level entries follow up to HF_PAIR_MAX_SUB
// second-level tables of 1 << HF_PAIR_W2 entries, one per slow index, over the
// next HF_PAIR_W2 bits of the window: entries of the same form for the w1 +
// HF_PAIR_W2 bits known there.  A slow first-level entry that has such a
// table carries HF_PAIR_SUB and the table's number << 8 instead of a symbol.
#define HF_PAIR_SLOW 0x80000000u
#define HF_PAIR_UNRESOLVED 0x40000000u
#define HF_PAIR_SUB 0x20000000u
#define HF_PAIR_W2 6u
#define HF_PAIR_MAX_SUB 32u
bool build_pair_lut(const DecodeLut &lut, double max_slow, const uint64_t *keys, const uint8_t *syms,
                    const uint8_t *klens, uint64_t m, uint32_t max_sub, std::vector<uint32_t> *pair);
// Layout of the decoder's LDS image of the stream.  Lane l of a wave starts
// reading near word l * mean_bits / 32 (chunk l of the segment), and 32 lanes
// are served per LDS cycle, one word per bank (word index mod 32): chunk sizes
// that are close to a multiple of a power of two put many lanes on one bank.
// The padded image leaves one slot unused after every 32 words, which moves
// those lanes apart but costs two instructions per word read.  Returns true
// when a simulation of the start positions (random walk with the chunk
// length's mean and standard deviation, fixed seed) expects clearly fewer
// serialised accesses with the padding.
bool prefer_padded_image(double mean_bits, double sd_bits);

} // namespace hf
#endif
