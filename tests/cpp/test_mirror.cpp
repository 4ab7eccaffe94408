// Scenarios of the reference crate's unit tests, replayed through the C++
// mirror (include/hfgpu.hpp) and therefore through libhfgpu.so on the GPU.
// Each case names the reference test it restates.  Run by pytest (-m gpu).
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <unistd.h>

#include "hfgpu.hpp"

using namespace rscompress_huffman;
using huffman::Huffman;
using huffman::encode::calculate_length;
using huffman::encode::Encoder;

static int g_failed = 0;
#define CHECK(cond)                                                                              \
    do {                                                                                         \
        if (!(cond)) {                                                                           \
            std::printf("    CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__);            \
            throw std::runtime_error("check failed");                                            \
        }                                                                                        \
    } while (0)

static void run(const char *name, const std::function<void()> &fn)
{
    try {
        fn();
        std::printf("ok   %s\n", name);
    } catch (const std::exception &e) {
        std::printf("FAIL %s: %s\n", name, e.what());
        ++g_failed;
    }
    std::fflush(stdout);
}

static const std::vector<u8> WORDS = {177, 112, 84, 143, 148, 195, 165, 206, 34, 10};

// the "code = value" model of encode_numbers / decode_numbers / serialisation_roundtrip
static Huffman numbers_model()
{
    std::array<usize, 256> codewords{}, length{};
    for (u8 w : WORDS) {
        codewords[w] = w;
        length[w] = calculate_length(w);
    }
    return Huffman(codewords, length);
}

static std::vector<u8> bytes_of(const char *s) { return std::vector<u8>(s, s + std::strlen(s)); }

int main()
{
    // src/lib.rs:108-131 test_u64_to_bytes
    run("test_u64_to_bytes", [] {
        CHECK((u64_to_bytes(341) == std::array<u8, 8>{0, 0, 0, 0, 0, 0, 1, 85}));
        CHECK((u64_to_bytes(1) == std::array<u8, 8>{0, 0, 0, 0, 0, 0, 0, 1}));
        CHECK((u64_to_bytes(32425) == std::array<u8, 8>{0, 0, 0, 0, 0, 0, 126, 169}));
        CHECK((u64_to_bytes(23534134) == std::array<u8, 8>{0, 0, 0, 0, 1, 103, 26, 54}));
        CHECK((u64_to_bytes(3234159383273ull) == std::array<u8, 8>{0, 0, 2, 241, 2, 235, 210, 233}));
        CHECK((u64_to_bytes(534457654) == std::array<u8, 8>{0, 0, 0, 0, 31, 219, 45, 54}));
        CHECK((u64_to_bytes(8273839836383ull) == std::array<u8, 8>{0, 0, 7, 134, 103, 72, 204, 223}));
    });
    // src/lib.rs:133-141 test_u64_to_bytes_roundtrip
    run("test_u64_to_bytes_roundtrip", [] {
        for (std::uint64_t v : {341ull, 1ull, 32425ull, 23534134ull, 3234159383273ull, 534457654ull,
                                8273839836383ull})
            CHECK(bytes_to_u64(u64_to_bytes(v).data()) == v);
    });
    // src/stats.rs:49-60 testing_histogram
    run("testing_histogram", [] {
        const std::vector<u8> data = {3, 32, 34, 34, 34, 2, 0};
        SliceReader r(data);
        auto h = stats::generate_histogram(r);
        CHECK(h[233] == 0 && h[3] == 1 && h[32] == 1 && h[2] == 1 && h[0] == 1 && h[34] == 3);
    });
    // src/huffman/mod.rs:213-230 test_extended_codewords_sorted_input
    run("test_extended_codewords_sorted_input", [] {
        auto cl = huffman::generate_extended_codewords({20, 17, 6, 3, 2, 2, 2, 1, 1, 1});
        const auto &c = cl.first;
        CHECK(c[0] == 0 && c[1] == 2 && c[2] == 12 && c[3] == 26);
        for (int i = 4; i <= 6; ++i)
            CHECK(c[i] >= 27 && c[i] <= 30);
        for (int i = 7; i <= 9; ++i)
            CHECK(c[i] >= 30 && c[i] <= 63);
    });
    // src/huffman/mod.rs:232-247 test_codeword_lengths_inplace
    run("test_codeword_lengths_inplace", [] {
        std::vector<usize> a = {10, 6, 2, 1, 1, 1};
        huffman::calculate_codeword_length_inplace(a);
        CHECK((a == std::vector<usize>{1, 2, 4, 4, 4, 4}));
        std::vector<usize> b = {20, 17, 6, 3, 2, 2, 2, 1, 1, 1};
        huffman::calculate_codeword_length_inplace(b);
        CHECK((b == std::vector<usize>{1, 2, 4, 5, 5, 5, 5, 5, 6, 6}));
        std::vector<usize> c = {99, 99, 99, 1, 1, 1};
        huffman::calculate_codeword_length_inplace(c);
        CHECK((c == std::vector<usize>{2, 2, 2, 3, 4, 4}));
        std::vector<usize> d = {8, 7, 6, 5, 4, 3};
        huffman::calculate_codeword_length_inplace(d);
        CHECK((d == std::vector<usize>{2, 2, 3, 3, 3, 3}));
    });
    // src/huffman/encode.rs:179-186 binary_length
    run("binary_length", [] {
        CHECK(calculate_length(4) == 3 && calculate_length(16) == 5 && calculate_length(2) == 2);
        CHECK(calculate_length(0) == 1 && calculate_length(1) == 1);
    });
    // src/huffman/encode.rs:158-177 encode_numbers
    run("encode_numbers", [] {
        Huffman h = numbers_model();
        Encoder<VecWriter> enc(VecWriter{}, h);
        const usize output_bytes = enc.write(WORDS);
        enc.flush();
        CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 82, 62, 83, 14, 151, 58, 42}));
        CHECK(output_bytes == 9);
        CHECK(enc.readbytes == 10 && enc.writeout == 9);
    });
    // src/huffman/encode.rs:118-136: flush does not reset.  A second flush emits the
    // bytes of the 64-bit buffer again, a write after a flush continues in front of
    // them.  Expected bytes and counters: the reference's algorithm (its CPU
    // restatement) run on these very calls.
    run("flush_twice_and_write_after_flush", [] {
        Huffman h = numbers_model();
        {
            Encoder<VecWriter> enc(VecWriter{}, h);
            CHECK(enc.write(WORDS) == 9 && enc.writeout == 5);
            enc.flush();
            CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 82, 62, 83, 14, 151, 58, 42}));
            CHECK(enc.writeout == 9 && *enc.fillbits == 0);
            enc.flush();
            CHECK((enc.inner.get_ref() ==
                   std::vector<u8>{177, 225, 82, 62, 83, 14, 151, 58, 42, 14, 151, 58, 42}));
            CHECK(enc.writeout == 13 && *enc.fillbits == 0);
            CHECK(enc.write(WORDS.data(), 4) == 8 && enc.writeout == 18 && enc.readbytes == 14);
            enc.flush();
            CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 82, 62, 83, 14,  151, 58, 42, 14, 151,
                                                         58,  42,  14, 151, 58, 42, 177, 225, 82, 60}));
            CHECK(enc.writeout == 21 && *enc.fillbits == 2);
        }
        {
            // the stream ends inside a byte at the first flush
            Encoder<VecWriter> enc(VecWriter{}, h);
            CHECK(enc.write(WORDS.data(), 3) == 3 && enc.writeout == 0);
            enc.flush();
            CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 80}) && enc.writeout == 3);
            CHECK(*enc.fillbits == 2);
            enc.flush();
            CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 80, 177, 225, 80}) && enc.writeout == 6);
            CHECK(enc.write(WORDS.data() + 3, 7) == 9 && enc.writeout == 11);
            enc.flush();
            CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 80, 177, 225, 80, 177, 225, 82, 62, 83, 14,
                                                         151, 58, 42}));
            CHECK(enc.writeout == 15 && *enc.fillbits == 0);
            // chunks of one size no longer describe this stream: the gap array is dropped
            CHECK(!enc.gap().fits(10, 9));
        }
        {
            // without the schedule the bytes to repeat are not known: refused, not guessed
            Encoder<VecWriter> enc(VecWriter{}, h);
            enc.track_write_value = false;
            enc.write(WORDS);
            enc.flush();
            CHECK((enc.inner.get_ref() == std::vector<u8>{177, 225, 82, 62, 83, 14, 151, 58, 42}));
            bool refused = false;
            try {
                enc.flush();
            } catch (const IoError &) {
                refused = true;
            }
            CHECK(refused);
        }
    });
    // src/huffman/encode.rs:188-205 encode_stream
    run("encode_stream", [] {
        std::array<usize, 256> codewords{}, length{};
        codewords[0] = 0;
        codewords[1] = 3;
        codewords[2] = 342;
        for (int i = 0; i < 3; ++i)
            length[i] = calculate_length(codewords[i]);
        Huffman h(codewords, length);
        Encoder<VecWriter> enc(VecWriter{}, h);
        const u8 in[3] = {0, 1, 2};
        const usize output_bytes = enc.write(in, 3);
        enc.flush();
        CHECK((enc.inner.get_ref() == std::vector<u8>{117, 96}));
        CHECK(output_bytes == 2);
        CHECK(enc.fillbits.has_value());
    });
    // src/huffman/decode/mod.rs:240-258 decode_numbers
    run("decode_numbers", [] {
        Huffman h = numbers_model();
        Encoder<VecWriter> enc(VecWriter{}, h);
        enc.write(WORDS);
        enc.flush();
        auto decoded = huffman::decode::read(enc.inner.get_ref(), h, enc.readbytes);
        CHECK(decoded == WORDS);
        // same through the chunk-parallel kernel, driven by the encoder's gap array
        auto decoded_gap = huffman::decode::read(enc.inner.get_ref(), h, enc.readbytes, &enc.gap());
        CHECK(decoded_gap == WORDS);
    });
    // src/huffman/decode/mod.rs:260-278 decode_numbers_histogram_encoded
    run("decode_numbers_histogram_encoded", [] {
        std::array<usize, 256> histogram{};
        const usize w[10] = {20, 17, 6, 3, 2, 2, 2, 1, 1, 1};
        for (int i = 0; i < 10; ++i)
            histogram[i] = w[i];
        Huffman h = Huffman::from_histogram(histogram);
        Encoder<VecWriter> enc(VecWriter{}, h);
        const std::vector<u8> origin = {0, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 0, 7, 4, 9, 9, 0, 0, 0, 4, 0};
        enc.write(origin);
        enc.flush();
        CHECK(huffman::decode::read(enc.inner.get_ref(), h, enc.readbytes) == origin);
        huffman::decode::Decoder<SliceReader> dec(SliceReader(enc.inner.get_ref()), enc);
        std::vector<u8> streamed;
        CHECK(dec.read_to_end(streamed) == origin.size() && streamed == origin);
    });
    // src/huffman/decode/vault.rs:265-270 roundtrip_blockwise
    run("vault_roundtrip_blockwise", [] {
        const std::pair<const char *, std::size_t> cases[] = {
            {"This is a lovely text in a big world", 10}, {"aaafaaaaaaaaa", 5}, {"aaaaaa", 2}};
        for (auto &cs : cases) {
            const auto data = bytes_of(cs.first);
            Huffman h = Huffman::from_slice(data);
            Encoder<VecWriter> enc(VecWriter{}, h);
            enc.write(data);
            enc.flush();
            const std::vector<u8> encoded = enc.inner.get_ref();
            huffman::decode::vault::Decoder<SliceReader> dec(SliceReader(encoded), h, data.size());
            std::vector<u8> block(cs.second);
            std::size_t nbytes = 1, iteration = 0;
            while (nbytes > 0) {
                nbytes = dec.read(block.data(), block.size());
                CHECK(std::equal(block.begin(), block.begin() + nbytes, data.begin() + iteration));
                iteration += nbytes;
            }
            CHECK(iteration == data.size());
        }
    });
    // src/huffman/decode/vault.rs:272-277 roundtrip_at_once
    run("vault_roundtrip_at_once", [] {
        for (const char *s : {"This is a lovely text in a big world", "aaafaaaaaaaaa", "aaaaaa"}) {
            const auto data = bytes_of(s);
            Huffman h = Huffman::from_slice(data);
            Encoder<VecWriter> enc(VecWriter{}, h);
            enc.write(data);
            enc.flush();
            const std::vector<u8> encoded = enc.inner.get_ref();
            huffman::decode::vault::Decoder<SliceReader> dec(SliceReader(encoded), h, data.size());
            std::vector<u8> decoded;
            CHECK(dec.read_to_end(decoded) == data.size());
            CHECK(decoded == data);
        }
    });
    // src/huffman/header.rs:59-77 serialisation_roundtrip
    run("serialisation_roundtrip", [] {
        Huffman h = numbers_model();
        Encoder<VecWriter> enc(VecWriter{}, h);
        auto head = huffman::header::Header::from(enc);
        auto temp = head.to_binary();
        auto new_head = huffman::header::Header::from_binary(temp);
        CHECK(new_head == head);
        CHECK(temp.size() == 36 + 10 * head.btree.size());
    });

    // ---- beyond the reference's own tests

    // search_key_or_next_small_key (decode/mod.rs:157-165): predecessor semantics and its panic
    run("search_key_or_next_small_key", [] {
        Huffman h = numbers_model();
        auto bt = h.to_btreemap();
        auto first = *bt.begin();
        CHECK(huffman::decode::search_key_or_next_small_key(bt, first.first) == first.second);
        CHECK(huffman::decode::search_key_or_next_small_key(bt, ~0ull) == bt.rbegin()->second);
        bool panicked = false;
        try {
            if (first.first > 0)
                huffman::decode::search_key_or_next_small_key(bt, first.first - 1);
            else
                panicked = true;
        } catch (const Panic &) {
            panicked = true;
        }
        CHECK(panicked);
        auto lk = huffman::decode::prepare_lookup(bt);
        CHECK(lk.first.size() == bt.size() && std::is_sorted(lk.second.begin(), lk.second.end()));
    });
    // the example of the crate docs: write in BUF pieces, read back in 15-byte pieces
    run("piecewise_write_and_15_byte_reads", [] {
        auto data = stats::generate_random_byte_vector(0, 40, 300000,
                                                       std::vector<u8>(40, 1), /*seed=*/7);
        for (std::size_t i = 0; i < data.size(); i += 3)
            data[i] = 5; // skew
        Huffman h = Huffman::from_slice(data);
        Encoder<VecWriter> whole(VecWriter{}, h), pieces(VecWriter{}, h);
        whole.track_write_value = false;
        whole.write(data);
        whole.flush();
        usize sum = 0;
        for (std::size_t off = 0; off < data.size(); off += BUF)
            sum += pieces.write(data.data() + off, std::min(BUF, data.size() - off));
        const usize before_flush = pieces.writeout;
        pieces.flush();
        CHECK(pieces.inner.get_ref() == whole.inner.get_ref());
        CHECK(before_flush <= pieces.writeout && pieces.writeout == whole.inner.get_ref().size());
        CHECK(sum >= before_flush); // write() over-reports by design of the reference (encode.rs:114)
        huffman::decode::Decoder<SliceReader> dec(SliceReader(pieces.inner.get_ref()), pieces);
        std::vector<u8> got, buf(15);
        while (std::size_t k = dec.read(buf.data(), buf.size()))
            got.insert(got.end(), buf.begin(), buf.begin() + k);
        CHECK(got == data);
        // header-only decoder without gap array: sequential walk on the device
        auto hd = huffman::header::Header::from(pieces);
        huffman::decode::Decoder<SliceReader> dec2(hd, SliceReader(pieces.inner.get_ref()));
        std::vector<u8> got2;
        dec2.read_to_end(got2);
        CHECK(got2 == data);
    });
    // src/lib.rs:31-80 file container, and its inverse through the gap side file
    run("stream_compress_with_header_information", [] {
        char dir[] = "/tmp/hfgpu_mirror_XXXXXX";
        CHECK(mkdtemp(dir) != nullptr);
        const std::string src = std::string(dir) + "/in.bin", dst = std::string(dir) + "/out.pzhf",
                          back = std::string(dir) + "/back.bin";
        auto data = stats::generate_random_byte_vector(10, 90, 1 << 20, [] {
            std::vector<u8> w(80);
            for (int i = 0; i < 80; ++i)
                w[i] = (u8)(1 + 200 / (i + 1));
            return w;
        }(), 11);
        {
            FileWriter f(src);
            f.write(data.data(), data.size());
        }
        stream_compress_with_header_information(src, dst);
        std::vector<u8> file;
        {
            FileReader f(dst);
            std::vector<u8> b(1 << 16);
            while (std::size_t k = f.read(b.data(), b.size()))
                file.insert(file.end(), b.begin(), b.begin() + k);
        }
        CHECK(file.size() > 12 && std::memcmp(file.data(), "pzhf", 4) == 0);
        const std::uint64_t hlen = bytes_to_u64(file.data() + 4);
        auto hd = huffman::header::Header::from_binary(file.data() + 12, hlen);
        CHECK(hd.readbytes == data.size() && hd.magic == bytes_of("pzhf"));
        CHECK(hlen == 36 + 10 * hd.btree.size());
        // the stream after the header is what Encoder produces for the same model
        Huffman h = Huffman::from_slice(data);
        Encoder<VecWriter> enc(VecWriter{}, h);
        enc.track_write_value = false;
        enc.write(data);
        enc.flush();
        CHECK(std::vector<u8>(file.begin() + 12 + hlen, file.end()) == enc.inner.get_ref());
        stream_decompress_with_header_information(dst, back);
        std::vector<u8> round;
        {
            FileReader f(back);
            std::vector<u8> b(1 << 16);
            while (std::size_t k = f.read(b.data(), b.size()))
                round.insert(round.end(), b.begin(), b.begin() + k);
        }
        CHECK(round == data);
        std::remove(src.c_str());
        std::remove(dst.c_str());
        std::remove(gap_sidecar_path(dst).c_str());
        std::remove(back.c_str());
        rmdir(dir);
    });
    // Encoder::write's return value and `writeout` (encode.rs:91-116) are kept by
    // counters on the host; the device replay of the same schedule must agree
    // call by call, whatever the call sizes
    run("write_schedule_matches_device_replay", [] {
        auto data = stats::generate_random_byte_vector(0, 60, 200000, [] {
            std::vector<u8> w(60);
            for (int i = 0; i < 60; ++i)
                w[i] = (u8)(1 + 250 / (i + 1));
            return w;
        }(), 3);
        Huffman h = Huffman::from_slice(data);
        Encoder<VecWriter> enc(VecWriter{}, h);
        usize state[3] = {0, 64, 0};
        std::size_t off = 0, call = 0;
        const std::size_t sizes[] = {1, 7, 4096, 15, 65536, 3, 4096, 100000};
        while (off < data.size()) {
            const std::size_t n = std::min(sizes[call++ % 8], data.size() - off);
            const usize got = enc.write(data.data() + off, n);
            usize want = 0;
            Runtime::check(hf_encoder_write_value_host(Runtime::ctx(), data.data() + off, n,
                                                       h.codewords().data(), h.length().data(),
                                                       state, &want),
                           "device replay");
            CHECK(got == want);
            CHECK(enc.writeout == state[2]);
            off += n;
        }
        enc.flush();
        CHECK(enc.writeout == enc.inner.get_ref().size());
    });
    // a gap array is only used when it fits the stream; a Decoder fed pieces
    run("gap_array_validation", [] {
        auto data = stats::generate_random_byte_vector(0, 30, 100000, std::vector<u8>(30, 3), 5);
        for (std::size_t i = 0; i < data.size(); i += 2)
            data[i] = 1;
        Huffman h = Huffman::from_slice(data);
        Encoder<VecWriter> enc(VecWriter{}, h);
        enc.write(data);
        enc.flush();
        const auto &stream = enc.inner.get_ref();
        huffman::GapArray g = enc.gap();
        CHECK(g.fits(data.size(), stream.size()) && g.fits(data.size(), std::nullopt));
        CHECK(!g.fits(data.size() - 1, stream.size()) && !g.fits(data.size(), stream.size() + 1));
        huffman::GapArray bad = g;
        bad.chunk_bits[5] += 1; // no longer adds up to the next segment offset
        CHECK(!bad.fits(data.size(), std::nullopt));
        bad = g;
        bad.seg_off[0] = 8;
        CHECK(!bad.fits(data.size(), stream.size()));
        bad = g;
        bad.chunk_bits.pop_back();
        CHECK(!bad.fits(data.size(), stream.size()));
        // decode::read and Decoder ignore what does not fit and still decode correctly
        bad = g;
        bad.chunk_bits[7] ^= 3;
        CHECK(huffman::decode::read(stream, h, data.size(), &bad) == data);
        auto hd = huffman::header::Header::from(enc);
        huffman::decode::Decoder<SliceReader> dec(hd, SliceReader(stream), &bad, stream.size());
        CHECK(!dec.uses_gap_array());
        std::vector<u8> got;
        dec.read_to_end(got);
        CHECK(got == data);
        huffman::decode::Decoder<SliceReader> dec2(hd, SliceReader(stream), &g, stream.size());
        CHECK(dec2.uses_gap_array());
        got.clear();
        dec2.read_to_end(got);
        CHECK(got == data);
    });
    // Codelen > 64 is InvalidData (encode.rs:102-104)
    run("codelen_over_64_is_invalid_data", [] {
        std::array<usize, 256> codewords{}, length{};
        codewords[1] = 1;
        length[1] = 65;
        Huffman h(codewords, length);
        bool invalid = false;
        try {
            Encoder<VecWriter> enc(VecWriter{}, h);
            const u8 in[1] = {1};
            enc.write(in, 1);
            enc.flush();
        } catch (const IoError &e) {
            invalid = e.invalid_data;
        }
        CHECK(invalid);
    });

    std::printf("%s\n", g_failed ? "FAILED" : "ALL OK");
    return g_failed ? 1 : 0;
}
