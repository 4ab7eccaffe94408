/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the rscompress/huffman hot path,
 * written to be read side by side with the reference sources.  It exists so
 * that the GPU library can be checked bit for bit.  Nothing in the shipped
 * library links, loads or calls this file: only tests/, the smoke check and the
 * cpu_baseline leg of bench.py may use it, and only as the checker.
 *
 * Pinning: the reference cannot be compiled in this environment (no Rust
 * toolchain, network-fetched dependencies), so this restatement is pinned by
 * every known-answer test the reference's own unit tests contain
 * (tests/golden/reference_kats.json, checked by tests/test_oracle_golden.py).
 * Two results are NOT pinned by any reference fixture and are documented as
 * "parity unpinned":
 *   - the order of equal-count symbols after sort_unstable_by_key
 *     (src/huffman/mod.rs:138); the rule used here is "equal counts keep
 *     ascending symbol order";
 *   - the bincode byte layout of Header (src/huffman/header.rs:39-46), taken
 *     from bincode 1.x's documented default (little-endian fixed-width ints,
 *     u64 length prefixes).
 * Third-party arithmetic restated from its published behaviour:
 *   - succinct (git sujayakar/succinct-rs, branch rsdict, no rev pinned):
 *     RsDict::rank1(i) = number of set bits in positions [0, i), saturating at
 *     the total once i passes the end of the vector.
 *
 * All Rust `usize` values are uint64_t here.  Arithmetic follows the release
 * profile of the reference (wrapping adds, shift amounts masked to 6 bits).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ stats */

/* src/stats.rs:6-10 (update_histogram) and :13-28 (generate_histogram): one
 * `+= 1` per byte; the 4096-byte read granularity (src/lib.rs:23) does not
 * change the counts, so the whole slice is walked at once.  Same loop as
 * Huffman::from_slice, src/huffman/mod.rs:59-65. */
ORC_EXPORT void orc_histogram(const uint8_t *data, uint64_t n, uint64_t hist[256])
{
    memset(hist, 0, 256 * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; ++i)
        hist[data[i]] += 1;
}

/* ------------------------------------------------------------- code lengths */

/* src/huffman/mod.rs:71-125 calculate_codeword_length_inplace.
 * Input: weights in non-increasing order.  Output: code lengths in place.
 * Returns -1 for n == 0 (the reference underflows `len() - 1` and panics). */
ORC_EXPORT int orc_codeword_lengths_inplace(uint64_t *h, uint64_t n)
{
    if (n == 0)
        return -1;
    int32_t leaf = (int32_t)(n - 1);  /* mod.rs:72 */
    uint64_t root = n - 1;            /* mod.rs:73 */

    /* Phase 1, mod.rs:76-93 */
    for (uint64_t next = n - 1; next >= 1; --next) {
        if (leaf < 0 || ((root > next) && (h[root] < h[(uint64_t)leaf]))) {
            h[next] = h[root];
            h[root] = next;
            root -= 1;
        } else {
            h[next] = h[(uint64_t)leaf];
            leaf -= 1;
        }
        if (leaf < 0 || ((root > next) && (h[root] < h[(uint64_t)leaf]))) {
            h[next] += h[root];
            h[root] = next;
            root -= 1;
        } else {
            h[next] += h[(uint64_t)leaf];
            leaf -= 1;
        }
    }

    /* Phase 2, mod.rs:96-103 */
    if (n != 1)
        h[1] = 0;
    for (uint64_t next = 2; next < n; ++next)
        h[next] = h[h[next]] + 1;

    /* Phase 3, mod.rs:105-124 */
    uint64_t avail = 1, used = 0, depth = 0, next = 0;
    root = 1;
    while (avail > 0) {
        while (root < n && h[root] == depth) {
            used += 1;
            root += 1;
        }
        while (avail > used) {
            h[next] = depth;
            next += 1;
            avail -= 1;
        }
        avail = 2 * used;
        depth += 1;
        used = 0;
    }
    return 0;
}

/* src/huffman/mod.rs:127-141 sort_by_value: keep (index, count) pairs with
 * count > 0, order by descending count.  The reference uses
 * sort_unstable_by_key, whose order among equal counts is unspecified
 * (the reference's own test says so, mod.rs:222-229).  Rule fixed here:
 * equal counts keep ascending index order (insertion sort, stable).
 * Returns the number of pairs, or -1 if n > 256 (assert at mod.rs:128). */
ORC_EXPORT int64_t orc_sort_by_value(const uint64_t *store, uint64_t n,
                                     uint64_t *idx_out, uint64_t *cnt_out)
{
    if (n > 256)
        return -1;
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (store[i] > 0) {
            uint64_t j = m;
            while (j > 0 && cnt_out[j - 1] < store[i]) {
                cnt_out[j] = cnt_out[j - 1];
                idx_out[j] = idx_out[j - 1];
                --j;
            }
            cnt_out[j] = store[i];
            idx_out[j] = i;
            ++m;
        }
    }
    return (int64_t)m;
}

/* src/huffman/mod.rs:155-171 calculate_codewords_based_on_length.
 * lengths must be non-decreasing; max = last length.  Returns -1 when
 * n == 0 or a shift would reach 64 bits. */
ORC_EXPORT int orc_codewords_from_lengths(const uint64_t *lengths, uint64_t n,
                                          uint64_t *small, uint64_t *big)
{
    if (n == 0)
        return -1;
    uint64_t maxlen = lengths[n - 1];
    if (maxlen > 63)
        return -1;
    small[0] = 0;
    big[0] = 0;
    for (uint64_t i = 1; i < n; ++i) {
        big[i] = ((uint64_t)1 << (maxlen - lengths[i - 1])) + big[i - 1];
        small[i] = big[i] >> (maxlen - lengths[i]);
    }
    return 0;
}

/* src/huffman/mod.rs:181-206 generate_extended_codewords.
 * hist may be shorter than 256 (the reference's test passes 10 entries). */
ORC_EXPORT int orc_generate_extended_codewords(const uint64_t *hist, uint64_t n,
                                               uint64_t codes[256], uint64_t lens[256])
{
    uint64_t idx[256], w[256], small[256], big[256];
    int64_t m = orc_sort_by_value(hist, n, idx, w);
    if (m < 0)
        return -1;
    if (m == 0)
        return -2; /* empty input: the reference panics (mod.rs:72) */
    orc_codeword_lengths_inplace(w, (uint64_t)m);
    if (m == 1)
        w[0] = 1; /* mod.rs:187-189 */
    if (orc_codewords_from_lengths(w, (uint64_t)m, small, big) != 0)
        return -3;
    memset(codes, 0, 256 * sizeof(uint64_t));
    memset(lens, 0, 256 * sizeof(uint64_t));
    for (int64_t i = 0; i < m; ++i) {
        codes[idx[i]] = small[i];
        lens[idx[i]] = w[i];
    }
    return 0;
}

/* ------------------------------------------------------------------- Model */

/* src/huffman/mod.rs:20-26 Model::sentinel = maximum code length. */
ORC_EXPORT uint64_t orc_sentinel(const uint64_t lens[256])
{
    uint64_t s = 0;
    for (int i = 0; i < 256; ++i)
        if (lens[i] > s)
            s = lens[i];
    return s;
}

/* src/huffman/mod.rs:27-38 Model::to_btreemap: key = code << (sentinel - len)
 * for every symbol with len != 0; later symbols overwrite equal keys
 * (BTreeMap::insert); output sorted by key.  Returns the entry count. */
ORC_EXPORT uint64_t orc_to_btreemap(const uint64_t codes[256], const uint64_t lens[256],
                                    uint64_t *keys, uint8_t *syms, uint8_t *klens)
{
    uint64_t s = orc_sentinel(lens);
    uint64_t m = 0;
    for (int ix = 0; ix < 256; ++ix) {
        uint64_t k = lens[ix];
        if (k == 0)
            continue;
        uint64_t key = codes[ix] << ((s - k) & 63);
        uint64_t pos = 0;
        while (pos < m && keys[pos] < key)
            ++pos;
        if (pos < m && keys[pos] == key) {
            syms[pos] = (uint8_t)ix;
            klens[pos] = (uint8_t)k;
            continue;
        }
        for (uint64_t j = m; j > pos; --j) {
            keys[j] = keys[j - 1];
            syms[j] = syms[j - 1];
            klens[j] = klens[j - 1];
        }
        keys[pos] = key;
        syms[pos] = (uint8_t)ix;
        klens[pos] = (uint8_t)k;
        ++m;
    }
    return m;
}

/* src/huffman/encode.rs:140-149 calculate_length */
ORC_EXPORT uint64_t orc_calculate_length(uint64_t val)
{
    if (val <= 1)
        return 1;
    uint64_t size = 0;
    while (size < 64 && (val >> size) > 0)
        size += 1;
    return size;
}

/* ----------------------------------------------------------------- Encoder */

/* src/huffman/encode.rs:26-34 */
typedef struct {
    uint64_t buffer;
    uint64_t remaining_bits;
    int32_t fillbits; /* -1 = None */
    uint64_t readbytes;
    uint64_t writeout;
    const uint64_t *codes;
    const uint64_t *lens;
    uint8_t *inner; /* growable byte sink standing in for `inner: W` */
    uint64_t inner_len;
    uint64_t inner_cap;
} orc_encoder;

static void enc_sink(orc_encoder *e, const uint8_t *b, uint64_t n)
{
    if (e->inner_len + n > e->inner_cap) {
        uint64_t cap = e->inner_cap ? e->inner_cap * 2 : 4096;
        while (cap < e->inner_len + n)
            cap *= 2;
        e->inner = (uint8_t *)realloc(e->inner, cap);
        e->inner_cap = cap;
    }
    memcpy(e->inner + e->inner_len, b, n);
    e->inner_len += n;
}

/* encode.rs:39-49 */
ORC_EXPORT orc_encoder *orc_encoder_new(const uint64_t codes[256], const uint64_t lens[256])
{
    orc_encoder *e = (orc_encoder *)calloc(1, sizeof(orc_encoder));
    e->buffer = 0;
    e->remaining_bits = 64;
    e->fillbits = -1;
    e->codes = codes;
    e->lens = lens;
    return e;
}

ORC_EXPORT void orc_encoder_free(orc_encoder *e)
{
    if (e) {
        free(e->inner);
        free(e);
    }
}

/* encode.rs:54-56 plain_write */
ORC_EXPORT void orc_encoder_plain_write(orc_encoder *e, const uint8_t *b, uint64_t n)
{
    enc_sink(e, b, n);
}

/* encode.rs:57-66 put */
static uint64_t enc_put(orc_encoder *e)
{
    uint8_t output = (uint8_t)(e->buffer >> 56);
    enc_sink(e, &output, 1);
    e->writeout += 1;
    e->buffer <<= 8;
    e->remaining_bits += 8;
    return 1;
}

/* encode.rs:67-83 cleanup */
static uint64_t enc_cleanup(orc_encoder *e)
{
    uint8_t output[5] = {
        (uint8_t)(e->buffer >> 56),
        (uint8_t)((e->buffer & 0x00FF000000000000ull) >> 48),
        (uint8_t)((e->buffer & 0x0000FF0000000000ull) >> 40),
        (uint8_t)((e->buffer & 0x000000FF00000000ull) >> 32),
        (uint8_t)((e->buffer & 0x00000000FF000000ull) >> 24),
    };
    enc_sink(e, output, 5);
    e->writeout += 5;
    e->buffer <<= 40;
    e->remaining_bits += 40;
    return 5;
}

/* encode.rs:91-116 Write::write.  Returns the reference's return value, or
 * -1 for the InvalidData error (codelen > 64, encode.rs:102-104). */
ORC_EXPORT int64_t orc_encoder_write(orc_encoder *e, const uint8_t *buf, uint64_t n)
{
    uint64_t writeout = 0;
    for (uint64_t i = 0; i < n; ++i) {
        e->readbytes += 1;
        uint64_t code = e->codes[buf[i]];
        uint64_t codelen = e->lens[buf[i]];
        if (codelen > 64)
            return -1;
        while (codelen > e->remaining_bits)
            writeout += enc_put(e);
        e->remaining_bits -= codelen;
        /* encode.rs:84-87 update_buffer: `+=` and `<<` with release semantics */
        e->buffer += code << (e->remaining_bits & 63);
        if ((e->buffer & 0x0000000000FF0000ull) > 0)
            writeout += enc_cleanup(e);
    }
    uint64_t tz = e->buffer ? (uint64_t)__builtin_ctzll(e->buffer) : 64;
    return (int64_t)(writeout + 8 - tz / 8);
}

/* encode.rs:118-136 flush (state is not reset; a second flush re-emits). */
ORC_EXPORT void orc_encoder_flush(orc_encoder *e)
{
    uint8_t w[8];
    for (int i = 0; i < 8; ++i)
        w[i] = (uint8_t)(e->buffer >> (56 - 8 * i));
    uint64_t length = 8 - e->remaining_bits / 8;
    e->fillbits = (int32_t)(e->remaining_bits % 8);
    enc_sink(e, w, length);
    e->writeout += length;
}

ORC_EXPORT const uint8_t *orc_encoder_data(const orc_encoder *e) { return e->inner; }
ORC_EXPORT uint64_t orc_encoder_len(const orc_encoder *e) { return e->inner_len; }
ORC_EXPORT uint64_t orc_encoder_readbytes(const orc_encoder *e) { return e->readbytes; }
ORC_EXPORT uint64_t orc_encoder_writeout(const orc_encoder *e) { return e->writeout; }
ORC_EXPORT int32_t orc_encoder_fillbits(const orc_encoder *e) { return e->fillbits; }

/* Convenience: Huffman::new + Encoder::new + write + flush on one slice
 * (the call pattern of examples/decode.rs:44-46).  Returns bytes produced,
 * or -1 on InvalidData, -2 when out_cap is too small. */
ORC_EXPORT int64_t orc_encode_slice(const uint8_t *data, uint64_t n,
                                    const uint64_t codes[256], const uint64_t lens[256],
                                    uint8_t *out, uint64_t out_cap, int64_t *write_ret)
{
    orc_encoder *e = orc_encoder_new(codes, lens);
    int64_t r = orc_encoder_write(e, data, n);
    if (r < 0) {
        orc_encoder_free(e);
        return -1;
    }
    orc_encoder_flush(e);
    if (write_ret)
        *write_ret = r;
    int64_t len = (int64_t)e->inner_len;
    if (e->inner_len > out_cap) {
        orc_encoder_free(e);
        return -2;
    }
    memcpy(out, e->inner, e->inner_len);
    orc_encoder_free(e);
    return len;
}

/* ------------------------------------------------------------------ decode */

/* decode/mod.rs:140-155 prepare_lookup + the lookup
 * `pos = rank1(searchvalue + 1) - 1` (decode/mod.rs:92,189,221), equal to
 * search_key_or_next_small_key (decode/mod.rs:157-165): index of the largest
 * key <= searchvalue.  Returns -1 when no key qualifies (the reference
 * underflows and panics). */
static int64_t predecessor(const uint64_t *keys, uint64_t m, uint64_t v)
{
    int64_t lo = 0, hi = (int64_t)m; /* first index with key > v */
    while (lo < hi) {
        int64_t mid = (lo + hi) / 2;
        if (keys[mid] <= v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo - 1;
}

ORC_EXPORT int64_t orc_search_key_or_next_small_key(const uint64_t *keys, uint64_t m, uint64_t key)
{
    return predecessor(keys, m, key);
}

/* decode/mod.rs:167-231 read(data, model, goalsbyte).
 * Returns the number of symbols produced (the reference returns a Vec that can
 * be longer than goalsbyte when data carries more bytes than the stream
 * needs), -1 on a window below the smallest key, -2 when out_cap is
 * exhausted, -3 for an empty model. */
ORC_EXPORT int64_t orc_decode_read(const uint8_t *data, uint64_t n,
                                   const uint64_t codes[256], const uint64_t lens[256],
                                   uint64_t goalsbyte, uint8_t *out, uint64_t out_cap)
{
    uint64_t keys[256];
    uint8_t syms[256], klens[256];
    uint64_t m = orc_to_btreemap(codes, lens, keys, syms, klens);
    if (m == 0)
        return -3;
    uint64_t s = orc_sentinel(lens);
    uint64_t shift = 64 - s;
    uint64_t buffer = 0;
    uint64_t bits_left = 64;
    uint64_t writeout = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t val = data[i];
        if (bits_left >= 8) { /* mod.rs:178-185 */
            buffer += val << (bits_left - 8);
            bits_left -= 8;
            continue;
        }
        while ((64 - bits_left) >= s) { /* mod.rs:187-206 */
            uint64_t searchvalue = buffer >> shift;
            int64_t pos = predecessor(keys, m, searchvalue);
            if (pos < 0)
                return -1;
            if (writeout >= out_cap)
                return -2;
            out[writeout++] = syms[pos];
            buffer <<= klens[pos];
            bits_left += klens[pos];
        }
        if (bits_left < 8)
            return -4; /* debug_assert at mod.rs:207-210; release would underflow */
        buffer += val << (bits_left - 8);
        bits_left -= 8;
    }
    while (goalsbyte > writeout) { /* mod.rs:219-229 */
        uint64_t searchvalue = buffer >> shift;
        int64_t pos = predecessor(keys, m, searchvalue);
        if (pos < 0)
            return -1;
        if (writeout >= out_cap)
            return -2;
        out[writeout++] = syms[pos];
        buffer <<= klens[pos];
        bits_left += klens[pos];
    }
    return (int64_t)writeout;
}

/* decode/mod.rs:15-26 streaming Decoder<R>; `inner` is a byte slice + cursor. */
typedef struct {
    const uint8_t *inner;
    uint64_t inner_len;
    uint64_t inner_pos;
    uint64_t buffer;
    uint8_t bits_left_in_buffer;
    uint64_t keys[256];
    uint8_t syms[256], klens[256];
    uint64_t m;
    uint64_t sentinel;
    uint64_t writeout;
    uint64_t goalsbyte;
    uint8_t shift;
} orc_decoder;

/* decode/mod.rs:29-47 Decoder::new (from an encoder's model and readbytes)
 * and :48-66 Decoder::from_header (from a key table). */
ORC_EXPORT orc_decoder *orc_decoder_from_table(const uint8_t *inner, uint64_t inner_len,
                                               const uint64_t *keys, const uint8_t *syms,
                                               const uint8_t *klens, uint64_t m,
                                               uint64_t sentinel, uint64_t goalsbyte)
{
    orc_decoder *d = (orc_decoder *)calloc(1, sizeof(orc_decoder));
    d->inner = inner;
    d->inner_len = inner_len;
    d->bits_left_in_buffer = 64;
    memcpy(d->keys, keys, m * sizeof(uint64_t));
    memcpy(d->syms, syms, m);
    memcpy(d->klens, klens, m);
    d->m = m;
    d->sentinel = sentinel;
    d->goalsbyte = goalsbyte;
    d->shift = (uint8_t)(64 - sentinel);
    return d;
}

ORC_EXPORT orc_decoder *orc_decoder_new(const uint8_t *inner, uint64_t inner_len,
                                        const uint64_t codes[256], const uint64_t lens[256],
                                        uint64_t readbytes)
{
    uint64_t keys[256];
    uint8_t syms[256], klens[256];
    uint64_t m = orc_to_btreemap(codes, lens, keys, syms, klens);
    return orc_decoder_from_table(inner, inner_len, keys, syms, klens, m,
                                  orc_sentinel(lens), readbytes);
}

ORC_EXPORT void orc_decoder_free(orc_decoder *d) { free(d); }

/* decode/mod.rs:70-137 Read::read.  Returns bytes produced or -1 on a
 * window below the smallest key. */
ORC_EXPORT int64_t orc_decoder_read(orc_decoder *d, uint8_t *buf, uint64_t buflen)
{
    uint64_t nbytes = d->goalsbyte - d->writeout;
    if (buflen < nbytes)
        nbytes = buflen;
    uint64_t consumed = 0;
    while (d->inner_pos < d->inner_len) { /* mod.rs:73-74 */
        uint64_t val = d->inner[d->inner_pos++];
        if (d->bits_left_in_buffer >= 16) { /* mod.rs:76-86 */
            d->buffer += val << (d->bits_left_in_buffer - 8);
            d->bits_left_in_buffer -= 8;
            continue;
        }
        while ((uint64_t)(64 - d->bits_left_in_buffer - 8) >= d->sentinel && consumed < nbytes) {
            uint64_t searchvalue = d->buffer >> d->shift;
            int64_t pos = predecessor(d->keys, d->m, searchvalue);
            if (pos < 0)
                return -1;
            buf[consumed++] = d->syms[pos];
            d->writeout += 1;
            d->buffer <<= d->klens[pos];
            d->bits_left_in_buffer += d->klens[pos];
        }
        d->buffer += val << (d->bits_left_in_buffer - 8); /* mod.rs:110-112 */
        d->bits_left_in_buffer -= 8;
        if (consumed >= nbytes)
            return (int64_t)consumed;
    }
    while (consumed < nbytes) { /* mod.rs:125-135 */
        uint64_t searchvalue = d->buffer >> d->shift;
        int64_t pos = predecessor(d->keys, d->m, searchvalue);
        if (pos < 0)
            return -1;
        buf[consumed++] = d->syms[pos];
        d->writeout += 1;
        d->buffer <<= d->klens[pos];
        d->bits_left_in_buffer += d->klens[pos];
    }
    return (int64_t)consumed;
}

/* ---------------------------------------------------------------- container */

/* src/lib.rs:82-93 u64_to_bytes (big endian) and :95-102 bytes_to_u64 */
ORC_EXPORT void orc_u64_to_bytes(uint64_t num, uint8_t out[8])
{
    for (int i = 0; i < 8; ++i)
        out[i] = (uint8_t)(num >> (56 - 8 * i));
}

ORC_EXPORT uint64_t orc_bytes_to_u64(const uint8_t in[8])
{
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i)
        r += (uint64_t)in[i] << (56 - i * 8);
    return r;
}

static uint8_t *put_le64(uint8_t *p, uint64_t v)
{
    for (int i = 0; i < 8; ++i)
        p[i] = (uint8_t)(v >> (8 * i));
    return p + 8;
}

/* src/huffman/header.rs:9-15,39-43 Header::to_binary through bincode 1.x's
 * default options: Vec<u8> = u64-LE length + bytes; BTreeMap = u64-LE length
 * + (u64-LE key, u8, u8) per entry in key order; usize = u64-LE.
 * Layout is parity-unpinned (no byte-level fixture in the reference).
 * Returns the byte count (36 + 10*m); out must hold that many. */
ORC_EXPORT uint64_t orc_header_to_binary(const uint64_t *keys, const uint8_t *syms,
                                         const uint8_t *klens, uint64_t m, uint64_t sentinel,
                                         uint64_t readbytes, uint8_t *out)
{
    uint8_t *p = out;
    p = put_le64(p, 4);
    memcpy(p, "pzhf", 4); /* encode.rs:51-53 magic */
    p += 4;
    p = put_le64(p, m);
    for (uint64_t i = 0; i < m; ++i) {
        p = put_le64(p, keys[i]);
        *p++ = syms[i];
        *p++ = klens[i];
    }
    p = put_le64(p, sentinel);
    p = put_le64(p, readbytes);
    return (uint64_t)(p - out);
}

/* src/lib.rs:31-80 stream_compress_with_header_information on a slice:
 * "pzhf" || u64-BE header length || header || bitstream.
 * Returns total bytes, -1 on InvalidData, -2 capacity, -3 empty input. */
ORC_EXPORT int64_t orc_compress_with_header(const uint8_t *data, uint64_t n,
                                            uint8_t *out, uint64_t out_cap)
{
    uint64_t hist[256], codes[256], lens[256], keys[256];
    uint8_t syms[256], klens[256], hdr[36 + 10 * 256], be[8];
    orc_histogram(data, n, hist);
    if (orc_generate_extended_codewords(hist, 256, codes, lens) != 0)
        return -3;
    uint64_t m = orc_to_btreemap(codes, lens, keys, syms, klens);
    uint64_t hlen = orc_header_to_binary(keys, syms, klens, m, orc_sentinel(lens), n, hdr);
    orc_encoder *e = orc_encoder_new(codes, lens);
    orc_encoder_plain_write(e, (const uint8_t *)"pzhf", 4);
    orc_u64_to_bytes(hlen, be);
    orc_encoder_plain_write(e, be, 8);
    orc_encoder_plain_write(e, hdr, hlen);
    for (uint64_t off = 0; off < n; off += 4096) { /* lib.rs:68-77, BUF = 4096 */
        uint64_t k = n - off < 4096 ? n - off : 4096;
        if (orc_encoder_write(e, data + off, k) < 0) {
            orc_encoder_free(e);
            return -1;
        }
    }
    orc_encoder_flush(e);
    int64_t total = (int64_t)e->inner_len;
    if (e->inner_len > out_cap) {
        orc_encoder_free(e);
        return -2;
    }
    memcpy(out, e->inner, e->inner_len);
    orc_encoder_free(e);
    return total;
}
