// Randomised differential run of the C++ mirror's streaming adaptors
// (include/hfgpu.hpp) against the CPU oracle (oracle/oracle.c), built with
// HFGPU_PIECE_DIVISOR = 4096 so that inputs of a few MiB cross the adaptors'
// piece boundaries many times (64 KiB staged per encode, 16 Ki symbols or an
// 8 KiB window per decode, 16 KiB per histogram call, files above 256 KiB on
// the piecewise path).
//   fuzz_mirror [cases] [first seed] [scratch dir]
// Per case: Huffman::from_reader through a reader that returns short reads,
// Encoder::write in calls of random sizes (return values against the oracle's
// encoder when they are tracked), flush, the bytes / fillbits / readbytes /
// writeout against the oracle, Decoder::new (gap array) and
// Decoder::from_header (no gap array) through short reads and random read
// sizes, decode::read, and the file pair stream_compress / stream_decompress
// with and without the side file against the oracle's container.
// A search tool, not part of the test suite.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "hfgpu.hpp"

extern "C" {
struct orc_encoder;
int orc_generate_extended_codewords(const uint64_t *hist, uint64_t n, uint64_t codes[256],
                                    uint64_t lens[256]);
void orc_histogram(const uint8_t *data, uint64_t n, uint64_t hist[256]);
orc_encoder *orc_encoder_new(const uint64_t codes[256], const uint64_t lens[256]);
void orc_encoder_free(orc_encoder *e);
int64_t orc_encoder_write(orc_encoder *e, const uint8_t *buf, uint64_t n);
void orc_encoder_flush(orc_encoder *e);
const uint8_t *orc_encoder_data(const orc_encoder *e);
uint64_t orc_encoder_len(const orc_encoder *e);
uint64_t orc_encoder_readbytes(const orc_encoder *e);
uint64_t orc_encoder_writeout(const orc_encoder *e);
int32_t orc_encoder_fillbits(const orc_encoder *e);
int64_t orc_compress_with_header(const uint8_t *data, uint64_t n, uint8_t *out, uint64_t out_cap);
}

using namespace rscompress_huffman;
using Rng = std::mt19937_64;

static std::string g_tag;
#define CHECK(cond, what)                                                                     \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            std::fprintf(stderr, "FAILED %s: %s (%s:%d)\n", g_tag.c_str(), what, __FILE__,    \
                         __LINE__);                                                           \
            std::exit(1);                                                                     \
        }                                                                                     \
    } while (0)

static uint64_t uniform(Rng &r, uint64_t lo, uint64_t hi) // inclusive
{
    return lo + r() % (hi - lo + 1);
}
static uint64_t log_uniform(Rng &r, uint64_t hi)
{
    const double u = (double)(r() >> 11) / 9007199254740992.0;
    return (uint64_t)std::exp(u * std::log((double)hi + 1.0));
}

// A reader over a slice that hands out short reads of random sizes.
struct ShortReader {
    const u8 *p;
    std::size_t n, pos = 0;
    Rng rng;
    std::size_t cap;
    ShortReader(const std::vector<u8> &v, uint64_t seed, std::size_t max_read)
        : p(v.data()), n(v.size()), rng(seed), cap(max_read)
    {
    }
    std::size_t read(u8 *dst, std::size_t len)
    {
        std::size_t k = std::min(len, n - pos);
        if (k > 1)
            k = std::min<std::size_t>(k, 1 + rng() % cap);
        std::memcpy(dst, p + pos, k);
        pos += k;
        return k;
    }
    void rewind() { pos = 0; }
};

static std::vector<u8> random_data(Rng &r, std::size_t n)
{
    static const int alphabets[] = {1, 2, 3, 9, 40, 256};
    const int a = alphabets[r() % 6];
    const double s = (double)(r() % 250) / 100.0;
    std::vector<double> w(a);
    for (int k = 0; k < a; ++k)
        w[k] = std::pow(k + 1.0, -s);
    std::discrete_distribution<int> d(w.begin(), w.end());
    std::vector<u8> perm(256);
    for (int i = 0; i < 256; ++i)
        perm[i] = (u8)i;
    std::shuffle(perm.begin(), perm.end(), r);
    std::vector<u8> out(n);
    if (r() % 4 == 0) { // runs of one symbol: bit density changes along the stream
        std::size_t i = 0;
        while (i < n) {
            const u8 sym = perm[d(r)];
            const std::size_t run = std::min<std::size_t>(n - i, 1 + log_uniform(r, 50000));
            std::fill(out.begin() + (std::ptrdiff_t)i, out.begin() + (std::ptrdiff_t)(i + run), sym);
            i += run;
        }
    } else {
        for (auto &b : out)
            b = perm[d(r)];
    }
    return out;
}

static std::vector<u8> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<u8> v((std::size_t)f.tellg());
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)v.size());
    return v;
}

static void run_case(uint64_t seed, const std::string &dir)
{
    Rng r(seed);
    std::size_t n = r() % 16 == 0 ? uniform(r, 1, 300) : 1 + log_uniform(r, 3u << 20);
    if (r() % 8 == 0) // whole segments, whole staging buffers: nothing left for the last piece
        n = (std::size_t)uniform(r, 1, 40) * (r() % 2 ? 16384 : 65536);
    const auto data = random_data(r, n);
    g_tag = "seed " + std::to_string(seed) + " n=" + std::to_string(n);

    // oracle side
    uint64_t ohist[256], ocodes[256], olens[256];
    orc_histogram(data.data(), n, ohist);
    CHECK(orc_generate_extended_codewords(ohist, 256, ocodes, olens) == 0, "oracle codebook");
    orc_encoder *oe = orc_encoder_new(ocodes, olens);

    // histogram + codebook through a reader with short reads
    ShortReader sr(data, seed * 3 + 1, 1 + log_uniform(r, 100000));
    const auto h = huffman::Huffman::from_reader(sr);
    for (int s = 0; s < 256; ++s)
        CHECK(h.codewords()[s] == ocodes[s] && h.length()[s] == olens[s], "codebook");

    // Encoder::write in calls of random sizes
    huffman::encode::Encoder<VecWriter> enc(VecWriter{}, h);
    enc.track_write_value = r() % 2 == 0;
    const std::size_t max_call = r() % 3 == 0 ? 4096 : 1 + log_uniform(r, 400000);
    std::size_t off = 0;
    while (off < n) {
        const std::size_t k = std::min<std::size_t>(n - off, r() % 3 == 0 ? max_call : 1 + r() % max_call);
        const usize ret = enc.write(data.data() + off, k);
        const int64_t oret = orc_encoder_write(oe, data.data() + off, k);
        if (enc.track_write_value) {
            CHECK((int64_t)ret == oret, "write() return value");
            CHECK(enc.writeout == orc_encoder_writeout(oe), "writeout between calls");
        }
        off += k;
        CHECK(enc.readbytes == orc_encoder_readbytes(oe), "readbytes");
    }
    enc.flush();
    orc_encoder_flush(oe);
    const auto &stream = enc.inner.get_ref();
    CHECK(stream.size() == orc_encoder_len(oe), "stream length");
    CHECK(std::memcmp(stream.data(), orc_encoder_data(oe), stream.size()) == 0, "stream bytes");
    CHECK(enc.fillbits && (int32_t)*enc.fillbits == orc_encoder_fillbits(oe), "fillbits");
    CHECK(enc.writeout == orc_encoder_writeout(oe), "writeout after flush");
    CHECK(enc.gap().fits(n, stream.size()), "gap array of the encoder fits its stream");
    orc_encoder_free(oe);

    // flush() in the middle, twice in a row, and write() after it: the reference does not
    // reset, its buffer's bytes go out again (src/huffman/encode.rs:118-136)
    if (n <= (1u << 20) || r() % 4 == 0) {
        orc_encoder *of = orc_encoder_new(ocodes, olens);
        huffman::encode::Encoder<VecWriter> ef(VecWriter{}, h);
        const std::size_t max_piece = 1 + log_uniform(r, 200000);
        std::size_t at = 0;
        int flushes = 0;
        while (at < n || flushes < 2) {
            if (at < n && r() % 3) {
                const std::size_t k = std::min<std::size_t>(n - at, 1 + r() % max_piece);
                const usize ret = ef.write(data.data() + at, k);
                CHECK((int64_t)ret == orc_encoder_write(of, data.data() + at, k), "write() value around flushes");
                at += k;
            } else {
                ef.flush();
                orc_encoder_flush(of);
                ++flushes;
                CHECK(ef.fillbits && (int32_t)*ef.fillbits == orc_encoder_fillbits(of), "fillbits of a repeated flush");
            }
            CHECK(ef.writeout == orc_encoder_writeout(of), "writeout around flushes");
            const auto &got = ef.inner.get_ref();
            if (flushes && (at == n || r() % 4 == 0)) // (what is still staged is not out yet)
                if (at == n || got.size() == orc_encoder_len(of))
                    CHECK(got.size() <= orc_encoder_len(of) &&
                              std::memcmp(got.data(), orc_encoder_data(of), got.size()) == 0,
                          "bytes around flushes");
        }
        ef.flush();
        orc_encoder_flush(of);
        const auto &got = ef.inner.get_ref();
        CHECK(got.size() == orc_encoder_len(of) && std::memcmp(got.data(), orc_encoder_data(of), got.size()) == 0,
              "bytes after repeated flushes");
        CHECK(ef.writeout == orc_encoder_writeout(of), "writeout after repeated flushes");
        orc_encoder_free(of);
    }

    // Decoder::new: with the encoder's gap array, short reads from the inner reader
    {
        huffman::decode::Decoder<ShortReader> dec(ShortReader(stream, seed * 5 + 2, 1 + log_uniform(r, 200000)), enc);
        CHECK(dec.uses_gap_array(), "Decoder::new uses the gap array");
        std::vector<u8> back(n + 8);
        const std::size_t max_read = r() % 3 == 0 ? 15 : 1 + log_uniform(r, 300000);
        std::size_t got = 0;
        while (true) {
            const std::size_t want = max_read == 15 ? 15 : 1 + r() % max_read;
            const std::size_t k = dec.read(back.data() + got, std::min(want, back.size() - got));
            if (k == 0)
                break;
            got += k;
        }
        CHECK(got == n && std::memcmp(back.data(), data.data(), n) == 0, "Decoder::new");
    }
    // Decoder::from_header: no gap array, the device finds the code boundaries per window
    {
        huffman::header::Header hd = huffman::header::Header::from(enc);
        const auto bin = hd.to_binary();
        const auto hd2 = huffman::header::Header::from_binary(bin);
        CHECK(hd2 == hd, "header round trip");
        huffman::decode::Decoder<ShortReader> dec(hd2, ShortReader(stream, seed * 7 + 3, 1 + log_uniform(r, 200000)));
        CHECK(!dec.uses_gap_array(), "Decoder::from_header without side data");
        std::vector<u8> back;
        if (r() % 2) {
            dec.read_to_end(back);
        } else {
            back.resize(n + 8);
            std::size_t got = 0;
            const std::size_t max_read = 1 + log_uniform(r, 300000);
            while (true) {
                const std::size_t k = dec.read(back.data() + got, std::min<std::size_t>(1 + r() % max_read, back.size() - got));
                if (k == 0)
                    break;
                got += k;
            }
            back.resize(got);
        }
        CHECK(back.size() == n && std::memcmp(back.data(), data.data(), n) == 0, "Decoder::from_header");
    }
    // decode::read on the slice, with and without the gap array
    {
        const auto a = huffman::decode::read(stream, h, n);
        CHECK(a == data, "decode::read");
        const auto b = huffman::decode::read(stream, h, n, &enc.gap());
        CHECK(b == data, "decode::read with gap array");
    }
    // the file pair
    if (r() % 2) {
        const std::string src = dir + "/src.bin", dst = dir + "/dst.pzhf", back = dir + "/back.bin";
        {
            std::ofstream f(src, std::ios::binary | std::ios::trunc);
            f.write(reinterpret_cast<const char *>(data.data()), (std::streamsize)n);
        }
        stream_compress_with_header_information(src, dst);
        std::vector<u8> want(n * 8 + 4096);
        const int64_t wn = orc_compress_with_header(data.data(), n, want.data(), want.size());
        CHECK(wn > 0, "oracle container");
        const auto file = read_file(dst);
        CHECK(file.size() == (std::size_t)wn && std::memcmp(file.data(), want.data(), file.size()) == 0,
              "container bytes");
        const bool drop_side = r() % 2;
        if (drop_side)
            std::remove(gap_sidecar_path(dst).c_str());
        stream_decompress_with_header_information(dst, back);
        CHECK(read_file(back) == data, drop_side ? "decompress without side file" : "decompress");
    }
}

int main(int argc, char **argv)
{
    const uint64_t cases = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 100;
    const uint64_t seed0 = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1;
    const std::string dir = argc > 3 ? argv[3] : "/tmp";
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t s = seed0; s < seed0 + cases; ++s) {
        try {
            run_case(s, dir);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "FAILED %s: exception %s\n", g_tag.c_str(), e.what());
            return 1;
        }
        if ((s - seed0) % 50 == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("[%6.1fs] %s ok\n", el, g_tag.c_str());
            std::fflush(stdout);
        }
    }
    std::printf("%llu cases passed\n", (unsigned long long)cases);
    return 0;
}
