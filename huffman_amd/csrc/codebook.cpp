// Host-side codebook construction and decode-table layout.
//
// The codebook is a <=256-element problem that the reference solves on the
// host in microseconds (src/huffman/mod.rs:71-206); it stays on the host here
// and only its result is staged into device memory.  The functions below give
// the same results as the reference's but are organised differently: code
// lengths come from an explicit two-queue Huffman merge followed by a
// leaf-depth census instead of the in-place three-phase array rewrite.
#include "hfgpu.h"
#include "hfgpu_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace hf {

// Lengths for weights sorted in non-increasing order.
// Equivalent to calculate_codeword_length_inplace (src/huffman/mod.rs:71-125):
//  * the two smallest available nodes are merged repeatedly; when a leaf and an
//    internal node weigh the same the leaf is taken first (the reference's
//    internal-node test is a strict `<`, mod.rs:77,85);
//  * only the NUMBER of leaves on each depth is kept; depths are then handed
//    out from the heaviest weight down (mod.rs:105-124), so the result is
//    non-decreasing even among equal weights.
int code_lengths_sorted(uint64_t *w, uint64_t n)
{
    if (n == 0)
        return HF_ERR_EMPTY;
    if (n == 1) {
        w[0] = 0;
        return HF_OK;
    }
    // Internal nodes are created in non-decreasing weight order, so a FIFO of
    // them plus a cursor over the leaves (lightest = last) is a priority queue.
    std::vector<uint64_t> iw(n - 1);       // weight of internal node k
    std::vector<uint32_t> parent(n - 1);   // parent of internal node k
    uint64_t made = 0, used = 0;           // internal nodes created / consumed
    int64_t leaf = (int64_t)n - 1;         // next unused leaf (lightest first)
    auto take = [&](uint32_t owner) -> uint64_t {
        bool use_internal = leaf < 0 || (used < made && iw[used] < w[leaf]);
        if (use_internal) {
            parent[used] = owner;
            return iw[used++];
        }
        return w[leaf--];
    };
    for (uint64_t k = 0; k + 1 < n; ++k) {
        uint64_t a = take((uint32_t)k);
        uint64_t b = take((uint32_t)k);
        iw[k] = a + b;
        made = k + 1;
    }
    // Depth of every internal node; the root is the last one created.
    std::vector<uint32_t> depth(n - 1);
    depth[n - 2] = 0;
    for (int64_t k = (int64_t)n - 3; k >= 0; --k)
        depth[k] = depth[parent[k]] + 1;
    // Census: a full binary tree level d holds 2 * (internal nodes on level
    // d-1) nodes; those that are not internal are leaves.
    std::vector<uint64_t> internal_on(n + 1, 0);
    for (uint64_t k = 0; k + 1 < n; ++k)
        internal_on[depth[k]] += 1;
    uint64_t next = 0, avail = 1;
    for (uint64_t d = 0; next < n; ++d) {
        uint64_t leaves = avail - internal_on[d];
        for (uint64_t i = 0; i < leaves && next < n; ++i)
            w[next++] = d;
        avail = 2 * internal_on[d];
        if (avail == 0)
            break;
    }
    return HF_OK;
}

// (index, count) of the non-zero entries, heaviest first.  The reference uses
// an unstable sort (src/huffman/mod.rs:138) whose order among equal counts is
// unspecified; the rule here is: equal counts stay in ascending index order.
int sort_nonzero_desc(const uint64_t *store, uint64_t n, uint64_t *idx, uint64_t *cnt, uint64_t *m)
{
    if (n > 256)
        return HF_ERR_BAD_ARG; // assert at mod.rs:128
    uint32_t order[256];
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (store[i] > 0)
            order[k++] = (uint32_t)i;
    std::stable_sort(order, order + k,
                     [&](uint32_t a, uint32_t b) { return store[a] > store[b]; });
    for (uint64_t i = 0; i < k; ++i) {
        idx[i] = order[i];
        cnt[i] = store[order[i]];
    }
    *m = k;
    return HF_OK;
}

// Canonical codes for non-decreasing lengths
// (src/huffman/mod.rs:155-171): the left-aligned ("big") code advances by one
// unit of the previous length; the right-aligned code is its top len bits.
int canonical_codes(const uint64_t *len, uint64_t n, uint64_t *small_codes, uint64_t *big_codes)
{
    if (n == 0)
        return HF_ERR_EMPTY;
    const uint64_t maxlen = len[n - 1];
    if (maxlen > 63)
        return HF_ERR_BAD_CODEBOOK;
    uint64_t left = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (len[i] > maxlen)
            return HF_ERR_BAD_CODEBOOK;
        if (i > 0)
            left += (uint64_t)1 << (maxlen - len[i - 1]);
        big_codes[i] = left;
        small_codes[i] = i == 0 ? 0 : left >> (maxlen - len[i]);
    }
    return HF_OK;
}

int build_codebook(const uint64_t *hist, uint64_t n, uint64_t codes[256], uint64_t lens[256])
{
    uint64_t idx[256], w[256], small_codes[256], big_codes[256], m = 0;
    int rc = sort_nonzero_desc(hist, n, idx, w, &m);
    if (rc != HF_OK)
        return rc;
    if (m == 0)
        return HF_ERR_EMPTY;
    rc = code_lengths_sorted(w, m);
    if (rc != HF_OK)
        return rc;
    if (m == 1)
        w[0] = 1; // a lone symbol still costs one bit (mod.rs:187-189)
    rc = canonical_codes(w, m, small_codes, big_codes);
    if (rc != HF_OK)
        return rc;
    std::memset(codes, 0, 256 * sizeof(uint64_t));
    std::memset(lens, 0, 256 * sizeof(uint64_t));
    for (uint64_t i = 0; i < m; ++i) {
        codes[idx[i]] = small_codes[i];
        lens[idx[i]] = w[i];
    }
    return HF_OK;
}

uint64_t sentinel_of(const uint64_t lens[256])
{
    uint64_t s = 0;
    for (int i = 0; i < 256; ++i)
        s = std::max(s, lens[i]);
    return s;
}

// Model::to_btreemap (src/huffman/mod.rs:27-38) as sorted parallel arrays.
// A later symbol replaces an earlier one with the same key, as
// BTreeMap::insert does.
int model_table(const uint64_t codes[256], const uint64_t lens[256], uint64_t keys[256],
                uint8_t syms[256], uint8_t klens[256], uint64_t *m)
{
    const uint64_t s = sentinel_of(lens);
    if (s > 64)
        return HF_ERR_CODELEN;
    struct Ent {
        uint64_t key;
        uint8_t sym, len;
    };
    std::vector<Ent> ents;
    for (int ix = 0; ix < 256; ++ix) {
        if (lens[ix] == 0)
            continue;
        uint64_t shift = s - lens[ix];
        uint64_t key = shift >= 64 ? codes[ix] : codes[ix] << shift;
        ents.push_back({key, (uint8_t)ix, (uint8_t)lens[ix]});
    }
    std::stable_sort(ents.begin(), ents.end(),
                     [](const Ent &a, const Ent &b) { return a.key < b.key; });
    uint64_t k = 0;
    for (size_t i = 0; i < ents.size(); ++i) {
        if (i + 1 < ents.size() && ents[i + 1].key == ents[i].key)
            continue; // the highest symbol index with this key wins
        keys[k] = ents[i].key;
        syms[k] = ents[i].sym;
        klens[k] = ents[i].len;
        ++k;
    }
    *m = k;
    return HF_OK;
}

// Index of the largest key <= v, or -1.  This is the reference's
// `rank1(v + 1) - 1` over the key bit vector (src/huffman/decode/mod.rs:92)
// and its BTree variant (:157-165).
static int64_t predecessor(const uint64_t *keys, uint64_t m, uint64_t v)
{
    const uint64_t *p = std::upper_bound(keys, keys + m, v);
    return (int64_t)(p - keys) - 1;
}

// Multi-level decode table.
//
// Level 1 is indexed by the top W1 bits of the bit window; every further level
// by the next W2 bits.  A slot whose whole window range has one predecessor
// key holds that key's (symbol, length) directly; a slot that straddles a key
// points to a sub-table; slots that cannot be resolved within 32 window bits
// (or once the pool is full) send the decoder to a binary search over the
// sorted keys.  Built by range-fill over predecessor semantics, so it answers
// exactly what the reference's rank lookup answers, prefix free or not.
int build_decode_lut(const uint64_t *keys, const uint8_t *syms, const uint8_t *klens, uint64_t m,
                     uint64_t sentinel, DecodeLut *lut)
{
    if (m == 0 || sentinel == 0)
        return HF_ERR_EMPTY;
    if (sentinel > HF_MAX_SENTINEL)
        return HF_ERR_UNSUPPORTED;
    const uint32_t s = (uint32_t)sentinel;
    const uint32_t w1 = std::min<uint32_t>(HF_LUT_W1, s);
    lut->w1 = w1;
    lut->sentinel = s;
    lut->pool.assign((size_t)1 << w1, 0);
    lut->n_sub = 0;
    lut->max_direct_len = 0;
    lut->needs_search = false;
    lut->has_escape = false;

    struct Job {
        size_t base;     // pool offset of the table
        uint64_t prefix; // window bits already consumed (value)
        uint32_t depth;  // number of window bits already consumed
        uint32_t width;  // index width of this table
    };
    std::vector<Job> jobs;
    jobs.push_back({0, 0, 0, w1});
    for (size_t j = 0; j < jobs.size(); ++j) {
        const Job job = jobs[j];
        // Only the first `sig` bits of the index are window bits; when the
        // window ends inside this table the rest of the index is don't-care.
        const uint32_t sig = std::min(job.width, s - job.depth);
        const uint32_t rem = s - job.depth - sig; // window bits below this slot
        // largest key <= the first slot's low end (or the first key), then swept forward
        int64_t cur_lo = predecessor(keys, m, (job.prefix << sig) << rem);
        if (cur_lo < 0)
            cur_lo = 0;
        int64_t cur_hi = cur_lo;
        for (uint32_t i = 0; i < (1u << job.width); ++i) {
            const uint64_t part = i >> (job.width - sig);
            const uint64_t lo = ((job.prefix << sig) | part) << rem;
            const uint64_t hi = lo + (((uint64_t)1 << rem) - 1);
            // lo and hi grow with i: the predecessors only ever move forward
            while (cur_lo + 1 < (int64_t)m && keys[cur_lo + 1] <= lo)
                ++cur_lo;
            if (cur_hi < cur_lo)
                cur_hi = cur_lo;
            while (cur_hi + 1 < (int64_t)m && keys[cur_hi + 1] <= hi)
                ++cur_hi;
            const int64_t plo = (cur_lo >= 0 && keys[cur_lo] <= lo) ? cur_lo : -1;
            const int64_t phi = (cur_hi >= 0 && keys[cur_hi] <= hi) ? cur_hi : -1;
            uint16_t e;
            if (phi < 0) {
                e = HF_LUT_INVALID;
            } else if (plo == phi && klens[plo] <= 32) {
                e = (uint16_t)HF_LUT_ENTRY(syms[plo], klens[plo]);
                lut->max_direct_len = std::max<uint32_t>(lut->max_direct_len, klens[plo]);
            } else if (plo != phi && job.depth + job.width + HF_LUT_W2 <= 32 &&
                       lut->n_sub < HF_LUT_MAX_SUB) {
                e = (uint16_t)(HF_LUT_ESC | (lut->n_sub << 8));
                jobs.push_back({lut->pool.size(), (job.prefix << sig) | part, job.depth + sig,
                                HF_LUT_W2});
                lut->pool.resize(lut->pool.size() + ((size_t)1 << HF_LUT_W2), 0);
                lut->n_sub += 1;
            } else {
                e = HF_LUT_SEARCH;
                lut->needs_search = true;
            }
            lut->has_escape = lut->has_escape || (e & HF_LUT_ESC);
            lut->pool[job.base + i] = e;
            // a direct entry repeats for every following slot that ends before
            // the next key: fill the run instead of deriving each slot
            if (!(e & HF_LUT_ESC)) {
                const uint32_t dc = job.width - sig;
                uint64_t part_end = (uint64_t)1 << sig;
                if (plo + 1 < (int64_t)m)
                    part_end = std::min(part_end, (keys[plo + 1] >> rem) - (job.prefix << sig));
                const uint64_t i_end = part_end << dc;
                for (uint64_t q = i + 1; q < i_end; ++q)
                    lut->pool[job.base + q] = e;
                if (i_end > i + 1)
                    i = (uint32_t)(i_end - 1);
            }
        }
    }
    return HF_OK;
}

bool is_complete_prefix_code(const uint64_t *keys, const uint8_t *klens, uint64_t m,
                             uint32_t sentinel)
{
    if (m == 0 || sentinel == 0 || sentinel > 64 || keys[0] != 0)
        return false;
    // with the keys (left-aligned codes) in ascending order, every key must
    // start exactly where the range that its predecessor's code covers ends,
    // and the last range must end with the key space
    for (uint64_t i = 0; i < m; ++i) {
        if (klens[i] == 0 || klens[i] > sentinel)
            return false;
        const uint64_t last = keys[i] + (((uint64_t)1 << (sentinel - klens[i])) - 1u);
        if (last < keys[i])
            return false;
        if (i + 1 < m ? keys[i + 1] != last + 1 || last + 1 == 0
                      : last != (sentinel == 64 ? ~(uint64_t)0 : ((uint64_t)1 << sentinel) - 1u))
            return false;
    }
    return true;
}

bool build_pair_lut(const DecodeLut &lut, double max_slow, const uint64_t *keys, const uint8_t *syms,
                    const uint8_t *klens, uint64_t m, uint32_t max_sub, std::vector<uint32_t> *pair)
{
    const uint32_t w = lut.w1;
    const uint32_t n = 1u << w;
    const uint32_t slow_limit = (uint32_t)(max_slow * (double)n);
    pair->assign(n, HF_PAIR_SLOW | HF_PAIR_UNRESOLVED);
    uint32_t slow = 0;
    std::vector<uint32_t> slow_at;
    for (uint32_t x = 0; x < n; ++x) {
        if (slow > slow_limit) { // most models are out after a few dozen indices
            pair->clear();
            return false;
        }
        const uint16_t e0 = lut.pool[x];
        const uint32_t l0 = HF_LUT_LEN(e0);
        if ((e0 & HF_LUT_ESC) || l0 == 0 || l0 > w) {
            ++slow;
            slow_at.push_back(x);
            continue;
        }
        // the bits behind the first code, zero filled: a second code that ends
        // inside the index is resolved by the first level whatever the fill
        const uint16_t e1 = lut.pool[(x << l0) & (n - 1)];
        const uint32_t l1 = HF_LUT_LEN(e1);
        if (l0 == w || (e1 & HF_LUT_ESC) || l1 == 0 || l0 + l1 > w) {
            (*pair)[x] = HF_PAIR_SLOW | (HF_LUT_SYM(e0) << 8) | l0; // first symbol only
            ++slow;
            slow_at.push_back(x);
            continue;
        }
        (*pair)[x] = (HF_LUT_SYM(e0) << 8) | (HF_LUT_SYM(e1) << 16) | (l0 + l1);
    }
    if (slow > slow_limit) {
        pair->clear();
        return false;
    }
    // Second level: the slow indices each get a table over the next
    // HF_PAIR_W2 bits of the window, filled the same way from the w +
    // HF_PAIR_W2 bits that are then known (the code at the head of a window
    // comes from the key search; it is the real one when it ends inside the
    // known bits, the table being a complete prefix code).  The first-level
    // entry then only names its table.
    const uint32_t s = lut.sentinel;
    auto head = [&](uint64_t win, uint32_t *sym, uint32_t *len) { // win: left-aligned 64 bits
        const int64_t k = predecessor(keys, m, win >> (64 - s));
        if (k < 0)
            return false;
        *sym = syms[k];
        *len = klens[k];
        return *len != 0;
    };
    const uint32_t known = w + HF_PAIR_W2;
    uint32_t n_sub = 0;
    for (uint32_t x : slow_at) {
        if (n_sub >= std::min(max_sub, HF_PAIR_MAX_SUB) || known > 32)
            break;
        const size_t base = pair->size();
        pair->resize(base + ((size_t)1 << HF_PAIR_W2), HF_PAIR_SLOW | HF_PAIR_UNRESOLVED);
        for (uint32_t y = 0; y < (1u << HF_PAIR_W2); ++y) {
            const uint64_t win = (uint64_t)((x << HF_PAIR_W2) | y) << (64 - known);
            uint32_t s0, l0, s1, l1;
            if (!head(win, &s0, &l0) || l0 > known)
                continue;
            if (l0 < known && head(win << l0, &s1, &l1) && l0 + l1 <= known)
                (*pair)[base + y] = (s0 << 8) | (s1 << 16) | (l0 + l1);
            else if (l0 <= 63)
                (*pair)[base + y] = HF_PAIR_SLOW | (s0 << 8) | l0;
        }
        (*pair)[x] = HF_PAIR_SLOW | HF_PAIR_SUB | (n_sub << 8);
        ++n_sub;
    }
    return true;
}

bool prefer_padded_image(double mean_bits, double sd_bits)
{
    if (!(mean_bits > 0))
        return false;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() { // uniform in [0, 1)
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(rng >> 11) * (1.0 / 9007199254740992.0);
    };
    double cost[2] = {0, 0};
    const int trials = 12;
    for (int t = 0; t < trials; ++t) {
        double start[32];
        double at = uni() * 32.0 * 1024.0; // bit phase of the group's first chunk
        for (int l = 0; l < 32; ++l) {
            start[l] = at;
            // sum of four uniforms: mean 2, variance 1/3
            const double g = (uni() + uni() + uni() + uni() - 2.0) * 1.7320508;
            at += std::max(1.0, mean_bits + sd_bits * g);
        }
        for (int padded = 0; padded < 2; ++padded) {
            // distinct words per bank (the same word is one access: broadcast);
            // the start words ascend, so a repeated word is the previous one
            int per_bank[32] = {0};
            uint32_t last_word[32];
            int worst = 1;
            for (int l = 0; l < 32; ++l) {
                uint32_t w = (uint32_t)(start[l] / 32.0);
                w = padded ? w + (w >> 5) : w;
                const uint32_t bank = w & 31u;
                if (per_bank[bank] == 0 || last_word[bank] != w) {
                    last_word[bank] = w;
                    worst = std::max(worst, ++per_bank[bank]);
                }
            }
            cost[padded] += worst;
        }
    }
    return cost[1] + 0.4 * trials < cost[0];
}

uint32_t pick_decode_window(const uint64_t *keys, const uint8_t *klens, uint64_t m,
                            uint32_t sentinel)
{
    if (m == 0 || sentinel == 0 || sentinel > 64 || 8 * sentinel <= 64)
        return 0; // the plain K = 64 / sentinel is at least 8 wide already
    if (!is_complete_prefix_code(keys, klens, m, sentinel))
        return 0;
    double d[65] = {0};
    double kraft = 0;
    int present[64], np = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const double w = klens[i] < 64 ? 1.0 / (double)((uint64_t)1 << klens[i]) : std::ldexp(1.0, -64);
        if (d[klens[i]] == 0)
            present[np++] = klens[i];
        d[klens[i]] += w;
        kraft += w;
    }
    for (double &x : d)
        x /= kraft;
    // P(sum of lengths = a) after 6, 7, 8 and 16 codes, everything above 64 in c[65]
    double c[66] = {0};
    c[0] = 1.0;
    const uint32_t widths[4] = {16u, 8u, 7u, 6u};
    // A chunk in ten thousand may hold a group that overflows its window: the
    // lane decodes its chunk again through global memory and its wave waits for
    // it, a few thousand instruction slots -- per wave (64 chunks) that is 0.6 %
    // of a few thousand, against the dozen instructions and three LDS reads
    // that every window rebuild costs every chunk.
    const double limit[4] = {1e-4, 1e-4, 1e-4, 1e-4};
    double over[4] = {1.0, 1.0, 1.0, 1.0};
    int lo = 0, hi = 0; // sums that can occur so far (below the overflow bin)
    int minlen = 64, maxl = 0;
    for (int q = 0; q < np; ++q)
        minlen = std::min(minlen, present[q]), maxl = std::max(maxl, present[q]);
    for (uint32_t i = 1; i <= 16; ++i) {
        double nc[66] = {0};
        nc[65] = c[65];
        for (int a = lo; a <= hi; ++a) {
            if (c[a] == 0)
                continue;
            for (int q = 0; q < np; ++q)
                nc[std::min(a + present[q], 65)] += c[a] * d[present[q]];
        }
        std::memcpy(c, nc, sizeof(c));
        lo = std::min(lo + minlen, 64), hi = std::min(hi + maxl, 64);
        for (int t = 0; t < 4; ++t)
            if (widths[t] == i)
                over[t] = c[65];
        // more codes overflow at least as often: once the most lenient test
        // fails for this many codes, nothing wider can pass
        if (i >= 6 && 1.0 - std::pow(1.0 - c[65], 64.0 / i) > 1e-4)
            break;
    }
    for (int t = 0; t < 4; ++t) {
        if (widths[t] * sentinel <= 64)
            continue;
        const double p_chunk = 1.0 - std::pow(1.0 - over[t], 64.0 / widths[t]);
        if (p_chunk <= limit[t])
            return widths[t];
    }
    return 0;
}

int pick_merge_width(const uint64_t lens[256], bool *speculative)
{
    double d[65] = {0}; // distribution of the code length of one symbol
    double kraft = 0;
    uint64_t maxlen = 0;
    for (int s = 0; s < 256; ++s) {
        if (lens[s] == 0 || lens[s] > 64)
            continue;
        const double w = lens[s] < 64 ? 1.0 / (double)((uint64_t)1 << lens[s]) : std::ldexp(1.0, -64);
        d[lens[s]] += w;
        kraft += w;
        maxlen = std::max(maxlen, lens[s]);
    }
    *speculative = maxlen > 32;
    if (kraft <= 0)
        return 1;
    int present[64], np = 0;
    for (int l = 1; l <= 64; ++l) {
        d[l] /= kraft;
        if (d[l] != 0)
            present[np++] = l;
    }
    for (int g : {4, 2}) {
        if (maxlen * g <= 32) {
            *speculative = false;
            return g;
        }
        // P(sum of g lengths > 32): convolve, lumping everything above 32
        double c[34] = {0};
        c[0] = 1.0;
        for (int i = 0; i < g; ++i) {
            double nc[34] = {0};
            for (int a = 0; a <= 33; ++a) {
                if (c[a] == 0)
                    continue;
                for (int q = 0; q < np; ++q)
                    nc[std::min(a + present[q], 33)] += c[a] * d[present[q]];
            }
            std::memcpy(c, nc, sizeof(c));
        }
        const double p_tile = 1.0 - std::pow(1.0 - c[33], 32.0 / g); // a lane's 32 symbols
        const double p_wave = 1.0 - std::pow(1.0 - p_tile, 64.0);
        if (p_wave <= 0.001) {
            *speculative = true;
            return g;
        }
    }
    return 1;
}

} // namespace hf

// ------------------------------------------------------------------ C ABI

extern "C" {

int hf_codeword_lengths_inplace(uint64_t *weights, uint64_t n)
{
    if (!weights && n)
        return HF_ERR_BAD_ARG;
    return hf::code_lengths_sorted(weights, n);
}

int hf_sort_by_value(const uint64_t *store, uint64_t n, uint64_t *idx, uint64_t *cnt, uint64_t *m)
{
    if (!store || !idx || !cnt || !m)
        return HF_ERR_BAD_ARG;
    return hf::sort_nonzero_desc(store, n, idx, cnt, m);
}

int hf_codewords_from_lengths(const uint64_t *lengths, uint64_t n, uint64_t *small_codes,
                              uint64_t *big_codes)
{
    if (!lengths || !small_codes || !big_codes)
        return HF_ERR_BAD_ARG;
    return hf::canonical_codes(lengths, n, small_codes, big_codes);
}

int hf_build_codebook(const uint64_t *hist, uint64_t n, uint64_t codes[256], uint64_t lens[256])
{
    if (!hist || !codes || !lens)
        return HF_ERR_BAD_ARG;
    return hf::build_codebook(hist, n, codes, lens);
}

uint64_t hf_model_sentinel(const uint64_t lens[256]) { return hf::sentinel_of(lens); }

int hf_model_to_table(const uint64_t codes[256], const uint64_t lens[256], uint64_t keys[256],
                      uint8_t syms[256], uint8_t klens[256], uint64_t *m)
{
    if (!codes || !lens || !keys || !syms || !klens || !m)
        return HF_ERR_BAD_ARG;
    return hf::model_table(codes, lens, keys, syms, klens, m);
}

int hf_shard_plan(const uint64_t *hists, const uint64_t lens[256], uint32_t world, uint64_t *bits,
                  uint64_t *start_bit, uint32_t *lead_bits, uint64_t *byte_first,
                  uint64_t *byte_last, uint64_t *total_bits)
{
    if (!hists || !lens || world == 0)
        return HF_ERR_BAD_ARG;
    uint64_t run = 0;
    for (uint32_t r = 0; r < world; ++r) {
        uint64_t b = 0;
        for (int s = 0; s < 256; ++s)
            b += hists[(size_t)r * 256 + s] * lens[s];
        if (bits)
            bits[r] = b;
        if (start_bit)
            start_bit[r] = run;
        if (lead_bits)
            lead_bits[r] = (uint32_t)(run % 8);
        if (byte_first)
            byte_first[r] = run / 8;
        if (byte_last)
            byte_last[r] = (run + b + 7) / 8;
        run += b;
    }
    if (total_bits)
        *total_bits = run;
    return HF_OK;
}

int hf_place_shard(uint8_t *whole, uint64_t whole_cap, const uint8_t *shard, uint64_t start_bit,
                   uint64_t shard_bits)
{
    if (!whole || (!shard && shard_bits))
        return HF_ERR_BAD_ARG;
    if (shard_bits == 0)
        return HF_OK;
    const uint64_t first = start_bit / 8, last = (start_bit + shard_bits + 7) / 8;
    if (last > whole_cap)
        return HF_ERR_CAPACITY;
    // the bytes shared with the predecessor and with the successor are OR-merged
    // (their bits are disjoint), everything in between is copied
    uint64_t i = 0, end = last - first;
    if (start_bit % 8) {
        whole[first] |= shard[0];
        i = 1;
    }
    if ((start_bit + shard_bits) % 8 && end > i) {
        whole[last - 1] |= shard[end - 1];
        end -= 1;
    }
    if (end > i)
        std::memcpy(whole + first + i, shard + i, end - i);
    return HF_OK;
}

uint64_t hf_calculate_length(uint64_t val)
{
    // src/huffman/encode.rs:140-149: bit length, with 0 and 1 both costing one bit
    return val <= 1 ? 1 : 64 - (uint64_t)__builtin_clzll(val);
}

} // extern "C"
