// Rates of the streaming adaptors the way the reference's callers drive them:
// Encoder::write in BUF = 4096-byte pieces (src/lib.rs:68-77) and Decoder::read
// in 15-byte pieces (examples/decode.rs:55-73).  usage: adaptor_rate [MiB] [--json]
// (--json: one more line with the four rates, for bench.py)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "hfgpu.hpp"

using namespace rscompress_huffman;
using huffman::Huffman;
using huffman::encode::Encoder;

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const std::size_t mib = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 256;
    const bool json = argc > 2 && !std::strcmp(argv[2], "--json");
    double rate[4] = {0, 0, 0, 0}; // write tracked, write untracked, read with gap array, read without
    const std::size_t n = mib << 20;
    std::vector<u8> data(n);
    std::uint64_t x = 88172645463325252ull;
    for (auto &b : data) { // skewed bytes, cheap to draw
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        const unsigned r = (unsigned)(x >> 40) & 0xffff;
        b = (u8)(r < 30000 ? 0 : r < 45000 ? 1 : r < 55000 ? 2 : (r >> 8));
    }
    Huffman h = Huffman::from_slice(data);
    for (int track = 1; track >= 0; --track) {
        Encoder<VecWriter> enc(VecWriter{}, h);
        enc.track_write_value = track != 0;
        enc.inner.buf.reserve(n);
        const double t0 = now();
        for (std::size_t off = 0; off < n; off += BUF)
            enc.write(data.data() + off, std::min(BUF, n - off));
        enc.flush();
        const double t1 = now();
        std::printf("Encoder::write in %zu-byte calls + flush, write() value %s: %.2f GB/s (%zu -> %zu bytes)\n",
                    BUF, track ? "tracked" : "not tracked", n / (t1 - t0) / 1e9, n,
                    enc.inner.get_ref().size());
        rate[track ? 0 : 1] = n / (t1 - t0) / 1e9;
        if (track)
            continue;
        for (int with_gap = 1; with_gap >= 0; --with_gap) {
            auto hd = huffman::header::Header::from(enc);
            huffman::decode::Decoder<SliceReader> dec(hd, SliceReader(enc.inner.get_ref()),
                                                      with_gap ? &enc.gap() : nullptr,
                                                      enc.inner.get_ref().size());
            std::vector<u8> out;
            out.reserve(n);
            u8 buf[15];
            const double t2 = now();
            while (std::size_t k = dec.read(buf, sizeof(buf)))
                out.insert(out.end(), buf, buf + k);
            const double t3 = now();
            std::printf("Decoder::read in 15-byte calls, %s gap array: %.2f GB/s%s\n",
                        with_gap ? "with the" : "without a", n / (t3 - t2) / 1e9,
                        out == data ? "" : "  MISMATCH");
            if (out != data)
                return 1;
            rate[with_gap ? 2 : 3] = n / (t3 - t2) / 1e9;
        }
    }
    if (json)
        std::printf("{\"unit\": \"GB/s\", \"bytes\": %zu, \"data\": \"skewed bytes in host memory\", "
                    "\"write_4096_tracked\": %.3f, \"write_4096_untracked\": %.3f, "
                    "\"read_15_with_gap_array\": %.3f, \"read_15_bare_stream\": %.3f}\n",
                    n, rate[0], rate[1], rate[2], rate[3]);
    return 0;
}
