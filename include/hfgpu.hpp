// Host-side C++ mirror of the public API of the rscompress-huffman crate, on
// top of the C ABI in hfgpu.h.  Same names, argument meaning and error
// behaviour as the reference (file:line cited per item); every byte of compute
// runs in libhfgpu.so on the GPU -- there is no CPU codec in here.
//
// Rust -> C++ conventions: `usize` is std::uint64_t; panics (`expect`,
// `unwrap`, index underflow) become Panic exceptions; io::Error becomes
// IoError (InvalidData keeps its kind); `impl Write` / `impl Read` are
// duck-typed template parameters:
//     Writer:  std::size_t write(const std::uint8_t*, std::size_t);  void flush();
//     Reader:  std::size_t read(std::uint8_t*, std::size_t);          (0 = end)
#ifndef HFGPU_HPP
#define HFGPU_HPP

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <istream>
#include <map>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hfgpu.h"

// The streaming adaptors move data in pieces (64 MiB of input per histogram
// call, 256 MiB staged per encode, 64 Mi symbols or a 32 MiB window per
// decode, files up to 1 GiB in one upload).  Test programs divide all of them
// by this factor so that inputs of a few MiB cross every piece boundary.
#ifndef HFGPU_PIECE_DIVISOR
#define HFGPU_PIECE_DIVISOR 1
#endif

namespace rscompress_huffman {

using usize = std::uint64_t;
using u8 = std::uint8_t;

constexpr std::size_t BUF = 4096; // src/lib.rs:23

struct Panic : std::runtime_error {
    int status;
    Panic(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};
struct IoError : std::runtime_error {
    bool invalid_data; // io::ErrorKind::InvalidData (src/huffman/encode.rs:102-104)
    IoError(bool inv, const std::string &what) : std::runtime_error(what), invalid_data(inv) {}
};

// One hf_ctx per process (device 0); created on first use.
class Runtime {
  public:
    static hf_ctx *ctx()
    {
        static Runtime rt;
        return rt.ctx_;
    }
    static void check(int status, const char *what)
    {
        if (status == HF_OK)
            return;
        std::string msg = std::string(what) + ": " + hf_strerror(status) + " (" +
                          hf_last_error(ctx()) + ")";
        if (status == HF_ERR_CODELEN)
            throw IoError(true, "Codelen > 64");
        throw Panic(status, msg);
    }

  private:
    Runtime()
    {
        if (hf_ctx_create(0, &ctx_) != HF_OK)
            throw Panic(HF_ERR_HIP, "hf_ctx_create failed: a GPU is required, there is no CPU path");
    }
    ~Runtime() { hf_ctx_destroy(ctx_); }
    hf_ctx *ctx_ = nullptr;
};

// ------------------------------------------------------------ src/model.rs

// trait Model (src/model.rs:3-7)
struct Model {
    virtual ~Model() = default;
    virtual std::pair<usize, usize> encode(u8 sym) const = 0;
    virtual usize sentinel() const = 0;
    virtual std::map<usize, std::pair<u8, u8>> to_btreemap() const = 0;
};

// ------------------------------------------------------------ src/stats.rs

namespace stats {

// generate_histogram (src/stats.rs:13-28): the reader is drained into a piece
// of PIECE bytes at a time; every piece is counted on the GPU and the 256
// counts are summed, so the host holds one piece whatever the input's size.
constexpr std::size_t HISTOGRAM_PIECE = (std::size_t(64) << 20) / HFGPU_PIECE_DIVISOR;
template <class Reader>
std::array<usize, 256> generate_histogram(Reader &reader)
{
    std::vector<u8> piece;
    std::array<usize, 256> hist{};
    bool end = false;
    while (!end) {
        // (grown as the reader delivers: a small input never allocates a whole piece)
        std::size_t have = 0;
        while (have < HISTOGRAM_PIECE) {
            if (piece.size() < have + BUF)
                piece.resize(std::min(HISTOGRAM_PIECE, std::max(piece.size() * 2, have + BUF)));
            const std::size_t n = reader.read(piece.data() + have, piece.size() - have);
            if (n == 0) {
                end = true;
                break;
            }
            have += n;
        }
        if (have == 0)
            break;
        std::array<usize, 256> part{};
        Runtime::check(hf_histogram_host(Runtime::ctx(), piece.data(), have, part.data()),
                       "generate_histogram");
        for (int i = 0; i < 256; ++i)
            hist[i] += part[i];
    }
    return hist;
}

// generate_random_byte_vector (src/stats.rs:36-42).  The reference draws from
// an unseeded thread_rng; a seed can be given here to make data reproducible.
inline std::vector<u8> generate_random_byte_vector(u8 min, u8 max, usize count,
                                                   const std::vector<u8> &weights,
                                                   std::uint64_t seed = std::random_device{}())
{
    if ((usize)(max - min) != weights.size())
        throw Panic(HF_ERR_BAD_ARG, "assertion failed: choices.len() == weights.len()");
    std::mt19937_64 rng(seed);
    std::discrete_distribution<int> dist(weights.begin(), weights.end());
    std::vector<u8> out(count);
    for (auto &b : out)
        b = (u8)(min + dist(rng));
    return out;
}

} // namespace stats

// ------------------------------------------------------ src/huffman/mod.rs

namespace huffman {

// calculate_codeword_length_inplace (src/huffman/mod.rs:71-125)
inline void calculate_codeword_length_inplace(std::vector<usize> &histogram)
{
    const int rc = hf_codeword_lengths_inplace(histogram.data(), histogram.size());
    if (rc != HF_OK)
        throw Panic(rc, "attempt to subtract with overflow (empty histogram)");
}

// sort_by_value (src/huffman/mod.rs:127-141)
inline std::vector<std::pair<usize, usize>> sort_by_value(const std::vector<usize> &store)
{
    usize idx[256], cnt[256], m = 0;
    const int rc = hf_sort_by_value(store.data(), store.size(), idx, cnt, &m);
    if (rc != HF_OK)
        throw Panic(rc, "assertion failed: store.len() <= 256");
    std::vector<std::pair<usize, usize>> out;
    for (usize i = 0; i < m; ++i)
        out.emplace_back(idx[i], cnt[i]);
    return out;
}

// extract_values (src/huffman/mod.rs:150-153)
inline std::vector<usize> extract_values(const std::vector<std::pair<usize, usize>> &store)
{
    std::vector<usize> v;
    for (auto &p : store)
        v.push_back(p.second);
    return v;
}

// calculate_codewords_based_on_length (src/huffman/mod.rs:155-171)
inline std::pair<std::vector<usize>, std::vector<usize>>
calculate_codewords_based_on_length(const std::vector<usize> &lengths)
{
    std::vector<usize> small_codes(lengths.size()), big_codes(lengths.size());
    const int rc = hf_codewords_from_lengths(lengths.data(), lengths.size(), small_codes.data(),
                                             big_codes.data());
    if (rc != HF_OK)
        throw Panic(rc, "calculate_codewords_based_on_length: bad lengths");
    return {small_codes, big_codes};
}

// generate_extended_codewords (src/huffman/mod.rs:181-206)
inline std::pair<std::array<usize, 256>, std::array<usize, 256>>
generate_extended_codewords(const std::vector<usize> &histogram)
{
    std::array<usize, 256> codes{}, lens{};
    const int rc = hf_build_codebook(histogram.data(), histogram.size(), codes.data(), lens.data());
    if (rc != HF_OK)
        throw Panic(rc, "generate_extended_codewords: empty or oversized histogram");
    return {codes, lens};
}

// struct Huffman (src/huffman/mod.rs:10-13) + impl Model (:15-39) + ctors (:44-66)
class Huffman : public Model {
  public:
    Huffman(const std::array<usize, 256> &codewords, const std::array<usize, 256> &length)
        : codewords_(codewords), length_(length)
    {
    }
    static Huffman from_histogram(const std::array<usize, 256> &histogram)
    {
        auto cl = generate_extended_codewords(
            std::vector<usize>(histogram.begin(), histogram.end()));
        return Huffman(cl.first, cl.second);
    }
    template <class SeekReader>
    static Huffman from_reader(SeekReader &reader)
    {
        auto hist = stats::generate_histogram(reader);
        reader.rewind(); // reader.seek(SeekFrom::Start(0)), mod.rs:54-56
        return from_histogram(hist);
    }
    static Huffman from_slice(const u8 *data, usize n)
    {
        std::array<usize, 256> hist{};
        Runtime::check(hf_histogram_host(Runtime::ctx(), data, n, hist.data()), "from_slice");
        return from_histogram(hist);
    }
    static Huffman from_slice(const std::vector<u8> &data)
    {
        return from_slice(data.data(), data.size());
    }

    std::pair<usize, usize> encode(u8 sym) const override
    {
        return {codewords_[sym], length_[sym]};
    }
    usize sentinel() const override { return hf_model_sentinel(length_.data()); }
    std::map<usize, std::pair<u8, u8>> to_btreemap() const override
    {
        usize keys[256], m = 0;
        u8 syms[256], klens[256];
        const int rc = hf_model_to_table(codewords_.data(), length_.data(), keys, syms, klens, &m);
        if (rc != HF_OK)
            throw Panic(rc, "to_btreemap");
        std::map<usize, std::pair<u8, u8>> bt;
        for (usize i = 0; i < m; ++i)
            bt[keys[i]] = {syms[i], klens[i]};
        return bt;
    }
    const std::array<usize, 256> &codewords() const { return codewords_; }
    const std::array<usize, 256> &length() const { return length_; }

  private:
    std::array<usize, 256> codewords_, length_;
};

// Device-resident tables of a Model (hf_model), built on demand.
class DeviceModel {
  public:
    explicit DeviceModel(const Model &m)
    {
        std::array<usize, 256> codes{}, lens{};
        for (int s = 0; s < 256; ++s) {
            auto cl = m.encode((u8)s);
            codes[s] = cl.first;
            lens[s] = cl.second;
        }
        Runtime::check(hf_model_create(Runtime::ctx(), codes.data(), lens.data(), &h_),
                       "Huffman model");
        codes_ = codes;
        lens_ = lens;
    }
    DeviceModel(const std::map<usize, std::pair<u8, u8>> &btree, usize sentinel)
    {
        std::vector<usize> keys;
        std::vector<u8> syms, klens;
        for (auto &kv : btree) {
            keys.push_back(kv.first);
            syms.push_back(kv.second.first);
            klens.push_back(kv.second.second);
        }
        Runtime::check(hf_model_from_table(Runtime::ctx(), keys.data(), syms.data(), klens.data(),
                                           keys.size(), sentinel, &h_),
                       "decode table");
    }
    ~DeviceModel() { hf_model_destroy(h_); }
    DeviceModel(const DeviceModel &) = delete;
    DeviceModel &operator=(const DeviceModel &) = delete;
    hf_model *handle() const { return h_; }
    const std::array<usize, 256> &codes() const { return codes_; }
    const std::array<usize, 256> &lens() const { return lens_; }

  private:
    hf_model *h_ = nullptr;
    std::array<usize, 256> codes_{}, lens_{};
};

// Side metadata of a stream written by Encoder: the gap array.
struct GapArray {
    std::uint32_t chunk_syms = 64;
    std::vector<std::uint16_t> chunk_bits;
    std::vector<usize> seg_off;
    usize symbols = 0; // symbols covered

    // Bit length of the stream the arrays describe.
    usize total_bits() const
    {
        if (seg_off.empty())
            return 0;
        usize bits = seg_off.back();
        for (std::size_t c = (seg_off.size() - 1) * HF_SEG_CHUNKS; c < chunk_bits.size(); ++c)
            bits += chunk_bits[c];
        return bits;
    }
    // Side metadata is only trusted when it is consistent in itself and fits
    // the stream it is used for: right symbol count and array sizes, the first
    // segment at bit 0, every segment offset equal to its predecessor plus the
    // chunk lengths in between and, when the stream's size is known, the last
    // bit inside the stream's last byte.  Anything else (a stale or foreign
    // side file) is ignored and the gap array is rebuilt from the stream.
    bool fits(usize readbytes, std::optional<usize> stream_bytes) const
    {
        usize nc = 0, ns = 0;
        if (symbols != readbytes || hf_gap_layout(readbytes, chunk_syms, &nc, &ns) != HF_OK)
            return false;
        if (chunk_bits.size() != nc || seg_off.size() != ns)
            return false;
        if (ns == 0)
            return true;
        if (seg_off[0] != 0)
            return false;
        usize run = 0;
        for (usize sg = 0; sg < ns; ++sg) {
            if (seg_off[sg] != run)
                return false;
            const usize c1 = std::min<usize>(nc, (sg + 1) * HF_SEG_CHUNKS);
            for (usize c = sg * HF_SEG_CHUNKS; c < c1; ++c)
                run += chunk_bits[c];
        }
        return !stream_bytes || (run + 7) / 8 == *stream_bytes;
    }
};

// -------------------------------------------------- src/huffman/encode.rs

namespace encode {

// calculate_length (src/huffman/encode.rs:140-149)
inline usize calculate_length(usize val) { return hf_calculate_length(val); }

// The byte-emission bookkeeping of Encoder::write and ::flush
// (src/huffman/encode.rs:57-136) without the inner writer: the reference's
// {buffer, remaining_bits, writeout} (:43-47), how many bytes `put` (:57-66)
// and `cleanup` (:67-83) hand on while the symbols of one call pass, what the
// call returns (:115), and the bytes a flush emits (:118-136).  The stream
// itself is produced on the GPU.  hf_encoder_write_value replays the same
// schedule on the device and is what the tests hold this against.
//
// advance() does not run the reference's loop symbol by symbol.  While a code
// still ends above bit 24 of the buffer neither `put` (the code fits) nor
// `cleanup` (bits 16-23 stay clear) can happen, and the bits themselves are
// not looked at until they are written out: such symbols only move
// remaining_bits, two at a time where a pair fits.  The others take the
// reference's step on a buffer whose low 24 bits are exact -- all that the
// cleanup test reads.  The buffer is made exact again at the end of the call
// by taking the symbols since the last cleanup once more, bits included.
struct WriteSchedule {
    usize buffer = 0, remaining_bits = 64, writeout = 0; // encode.rs:43-47

    void set_codes(const std::array<usize, 256> &codes, const std::array<usize, 256> &lens)
    {
        codes_ = &codes;
        lens_ = &lens;
        for (int a = 0; a < 256; ++a)
            len8_[a] = (u8)std::min<usize>(lens[a], 255);
        pair_.reset(new u8[65536]);
        for (int a = 0; a < 256; ++a) // (index = the two bytes as they lie in memory, either byte order)
            for (int b = 0; b < 256; ++b) {
                const u8 two[2] = {(u8)a, (u8)b};
                std::uint16_t at;
                std::memcpy(&at, two, 2);
                pair_[at] = (u8)std::min<usize>((usize)len8_[a] + len8_[b], 255);
            }
    }
    // One symbol as the reference takes it (encode.rs:93-112); returns the bytes emitted.
    usize step(u8 sym)
    {
        const usize code = (*codes_)[sym], len = (*lens_)[sym];
        usize emitted = 0;
        if (len > 64)
            throw IoError(true, "Codelen > 64"); // encode.rs:102-104
        while (len > remaining_bits) { // put: one byte out
            emitted += 1;
            buffer <<= 8;
            remaining_bits += 8;
        }
        remaining_bits -= len;
        buffer += code << (remaining_bits & 63);
        if (buffer & 0x0000000000FF0000ull) { // cleanup: five bytes out
            emitted += 5;
            buffer <<= 40;
            remaining_bits += 40;
        }
        return emitted;
    }
    // Returns the value Encoder::write returns for these symbols.
    usize advance(const u8 *buf, std::size_t n)
    {
        usize emitted = 0;
        std::size_t i = 0, exact_from = 0; // `buffer` lacks bits of the symbols [exact_from, i)
        usize exact_buffer = buffer, exact_remaining = remaining_bits;
        const u8 *len8 = len8_, *pair = pair_.get();
        while (i < n) {
            usize r = remaining_bits;
            while (i + 2 <= n) {
                std::uint16_t two;
                std::memcpy(&two, buf + i, 2);
                const usize l2 = pair[two];
                if (r < l2 + 24)
                    break;
                r -= l2;
                i += 2;
            }
            while (i < n && r >= (usize)len8[buf[i]] + 24) {
                r -= len8[buf[i]];
                ++i;
            }
            remaining_bits = r;
            if (i == n)
                break;
            const usize e = step(buf[i++]);
            emitted += e;
            if (e >= 5) { // a cleanup: everything that is left lay in the exact low bits
                exact_from = i;
                exact_buffer = buffer;
                exact_remaining = remaining_bits;
            }
        }
        const usize end_remaining = remaining_bits;
        buffer = exact_buffer;
        remaining_bits = exact_remaining;
        for (std::size_t k = exact_from; k < n; ++k)
            step(buf[k]);
        if (remaining_bits != end_remaining)
            throw Panic(HF_ERR_INTERNAL, "WriteSchedule: the two passes disagree");
        writeout += emitted;
        const usize tz = buffer ? (usize)__builtin_ctzll(buffer) : 64;
        return emitted + 8 - tz / 8;
    }
    // flush (encode.rs:118-136): the bytes of the buffer that hold bits; state unchanged
    // but for `writeout`.  Returns their number; out[0 .. that) receives them.
    usize flush(u8 out[8])
    {
        for (int k = 0; k < 8; ++k)
            out[k] = (u8)(buffer >> (56 - 8 * k));
        const usize length = 8 - remaining_bits / 8;
        writeout += length;
        return length;
    }

  private:
    const std::array<usize, 256> *codes_ = nullptr, *lens_ = nullptr;
    u8 len8_[256] = {};
    std::unique_ptr<u8[]> pair_;
};

// struct Encoder<'a, W: Write, M: Model> (src/huffman/encode.rs:26-34).
// write() stages the input; the GPU encodes when `flush()` is called (or when
// STAGE_LIMIT bytes are pending): the reference's progressive inner writes
// are not observable in the final stream.  The return value of write() and
// `writeout` between calls follow the reference's put/cleanup schedule
// exactly while `track_write_value` is set (WriteSchedule: on the host, no
// device round trip per call).  Otherwise write() returns the bytes it really
// handed to `inner` during the call (0 while staging) and `writeout` is exact
// again after flush().  Callers that ignore the return value, like the file
// compressor below, clear the flag and save the pass over the symbols.
//
// Like the reference's, flush() does not reset the encoder: the bytes that
// were still in the reference's 64-bit buffer are emitted by the flush and
// stay in the buffer, so a second flush() emits them again and a write()
// after a flush() continues in front of them (encode.rs:118-136).  Which bytes
// those are is part of the schedule: with `track_write_value` cleared at any
// write() so far (setting it again later does not bring the schedule back), an
// encoder that has been flushed refuses further use.
// The gap array describes the one continuous bit stream, not the bytes that
// were handed on twice; it is dropped when symbols follow a flush that ended
// inside a segment.
template <class W>
class Encoder {
  public:
    // bytes staged before the device encodes them: a file larger than this is
    // compressed piece by piece (pieces are whole segments, so the gap arrays
    // concatenate and only a shared boundary byte is carried over)
    static constexpr std::size_t STAGE_LIMIT = (std::size_t(256) << 20) / HFGPU_PIECE_DIVISOR;

    W inner;
    const Model *model;
    std::optional<u8> fillbits; // set by flush (encode.rs:130)
    usize readbytes = 0;
    usize writeout = 0;
    bool track_write_value = true;

    Encoder(W writer, const Model &m) : inner(std::move(writer)), model(&m) {} // encode.rs:39-49
    std::vector<u8> magic() const { return {'p', 'z', 'h', 'f'}; }             // encode.rs:51-53
    void plain_write(const u8 *bytes, std::size_t n) { write_all(bytes, n); }  // encode.rs:54-56
    void plain_write(const std::vector<u8> &b) { write_all(b.data(), b.size()); }

    // impl Write::write (encode.rs:91-116)
    usize write(const u8 *buf, std::size_t n)
    {
        ensure_device_model();
        refuse_after_untracked_flush();
        usize ret = 0;
        const bool replay = track_write_value && schedule_valid_;
        if (replay) {
            ret = schedule_.advance(buf, n);
            writeout = schedule_.writeout;
        } else if (n) {
            schedule_valid_ = false;
        }
        readbytes += n;
        // (one reservation instead of doubling: a growing vector would hold two
        // copies of the staged bytes at the moment it moves them)
        if (pending_.size() + n > (std::size_t(1) << 20) && pending_.capacity() < STAGE_LIMIT + n)
            pending_.reserve(STAGE_LIMIT + std::max<std::size_t>(n, BUF));
        pending_.insert(pending_.end(), buf, buf + n);
        const usize before = emitted_;
        if (pending_.size() >= STAGE_LIMIT)
            drain(false);
        return replay ? ret : emitted_ - before;
    }
    usize write(const std::vector<u8> &buf) { return write(buf.data(), buf.size()); }

    // impl Write::flush (encode.rs:118-136)
    void flush()
    {
        ensure_device_model();
        refuse_after_untracked_flush();
        drain(true);
        inner.flush();
    }

    const GapArray &gap() const { return gap_; }
    const DeviceModel &device_model()
    {
        ensure_device_model();
        return *dm_;
    }

  private:
    void ensure_device_model()
    {
        if (!dm_) {
            dm_ = std::make_unique<DeviceModel>(*model);
            schedule_.set_codes(dm_->codes(), dm_->lens());
        }
    }
    void refuse_after_untracked_flush() const
    {
        if (flushed_ && !schedule_valid_)
            throw IoError(false, "Encoder used again after flush(): which bytes the reference "
                                 "emits again is only known with track_write_value set");
    }
    void write_all(const u8 *b, std::size_t n)
    {
        std::size_t off = 0;
        while (off < n) {
            const std::size_t k = inner.write(b + off, n - off);
            if (k == 0)
                throw IoError(false, "failed to write whole buffer");
            off += k;
        }
    }
    // Encodes what is staged.  Whole bytes go to `inner`; the last partial byte is kept
    // and merged into the next batch (lead bits) -- after a flush, which emits it, too.
    void drain(bool final)
    {
        const std::uint32_t lead = (std::uint32_t)(total_bits_ % 8);
        const usize seg_syms = (usize)HF_SEG_CHUNKS * gap_.chunk_syms;
        const usize n = final ? pending_.size() : pending_.size() - pending_.size() % seg_syms;
        if (n == 0 && !final)
            return;
        // whole bytes the reference still had in its buffer at the last flush: out again
        if (!resend_.empty()) {
            write_all(resend_.data(), resend_.size());
            emitted_ += resend_.size();
            resend_.clear();
        }
        // (not value-initialised: only the bytes the stream really has are touched)
        const std::size_t out_cap = hf_encode_bound(dm_->handle(), n, lead) + 8;
        std::unique_ptr<u8[]> out(new u8[out_cap]);
        // symbols that follow a flush in the middle of a segment: the chunks would no
        // longer be of one size, so no gap array from here on (decoders then find the
        // code boundaries themselves)
        if (n && gap_.symbols % seg_syms != 0) {
            gap_ = GapArray{};
            gap_kept_ = false;
        }
        usize nch = 0, nsg = 0;
        if (gap_kept_)
            hf_gap_layout(n, gap_.chunk_syms, &nch, &nsg);
        std::vector<std::uint16_t> cb(nch);
        std::vector<usize> so(nsg);
        usize bits = 0;
        if (n)
            Runtime::check(hf_encode_host(Runtime::ctx(), dm_->handle(), pending_.data(), n, lead,
                                          out.get(), out_cap, gap_.chunk_syms,
                                          gap_kept_ ? cb.data() : nullptr,
                                          gap_kept_ ? so.data() : nullptr, &bits),
                           "Encoder::flush");
        else
            out[0] = 0; // a flush with nothing staged: only the partial byte, if there is one
        // gap array: segment offsets become absolute; batches are only cut at
        // multiples of a segment, so the arrays simply concatenate
        if (gap_kept_) {
            const usize base_bit = total_bits_ - lead;
            for (auto &v : so)
                v += base_bit;
            gap_.chunk_bits.insert(gap_.chunk_bits.end(), cb.begin(), cb.end());
            gap_.seg_off.insert(gap_.seg_off.end(), so.begin(), so.end());
            gap_.symbols += n;
        }
        usize nbytes = (lead + bits + 7) / 8;
        if (lead && nbytes)
            out[0] |= carry_;
        total_bits_ += bits;
        const bool partial = (total_bits_ % 8) != 0;
        if (partial && nbytes)
            carry_ = out[nbytes - 1];
        if (!final && partial && nbytes)
            --nbytes;
        write_all(out.get(), nbytes);
        emitted_ += nbytes;
        if (final) {
            fillbits = (u8)((8 - total_bits_ % 8) % 8);
            flushed_ = true;
            if (schedule_valid_) {
                // encode.rs:119-133: what the flush has just emitted from the reference's
                // buffer stays there; its whole bytes go out again before anything new
                u8 held[8];
                const usize length = schedule_.flush(held);
                resend_.assign(held, held + (length - (partial ? 1 : 0)));
                writeout = schedule_.writeout;
            } else {
                writeout = (total_bits_ + 7) / 8;
            }
        } else if (!track_write_value || !schedule_valid_) {
            writeout = (total_bits_ + 7) / 8;
        }
        pending_.erase(pending_.begin(), pending_.begin() + (std::ptrdiff_t)n);
    }

    std::unique_ptr<DeviceModel> dm_;
    std::vector<u8> pending_, resend_;
    GapArray gap_;
    usize total_bits_ = 0, emitted_ = 0;
    u8 carry_ = 0;
    bool flushed_ = false;
    bool gap_kept_ = true;
    bool schedule_valid_ = true; // every symbol so far went through schedule_
    WriteSchedule schedule_;
};

} // namespace encode

// --------------------------------------------------- src/huffman/header.rs

namespace header {

// struct Header (src/huffman/header.rs:9-15)
struct Header {
    std::vector<u8> magic;
    std::map<usize, std::pair<u8, u8>> btree;
    usize sentinel = 0;
    usize readbytes = 0;

    // impl From<&Encoder> (header.rs:22-31)
    template <class W>
    static Header from(const encode::Encoder<W> &enc)
    {
        Header h;
        h.magic = enc.magic();
        h.btree = enc.model->to_btreemap();
        h.sentinel = enc.model->sentinel();
        h.readbytes = enc.readbytes;
        return h;
    }
    // to_binary (header.rs:39-43): bincode 1.x default options -- u64
    // little-endian lengths and integers, map entries in key order.
    std::vector<u8> to_binary() const
    {
        std::vector<u8> out;
        auto le64 = [&](usize v) {
            for (int i = 0; i < 8; ++i)
                out.push_back((u8)(v >> (8 * i)));
        };
        le64(magic.size());
        out.insert(out.end(), magic.begin(), magic.end());
        le64(btree.size());
        for (auto &kv : btree) {
            le64(kv.first);
            out.push_back(kv.second.first);
            out.push_back(kv.second.second);
        }
        le64(sentinel);
        le64(readbytes);
        return out;
    }
    // from_binary (header.rs:44-46)
    static Header from_binary(const u8 *p, std::size_t n)
    {
        std::size_t off = 0;
        auto le64 = [&]() {
            if (off + 8 > n)
                throw Panic(HF_ERR_BAD_ARG, "called `Result::unwrap()` on an `Err` value: header");
            usize v = 0;
            for (int i = 0; i < 8; ++i)
                v |= (usize)p[off + i] << (8 * i);
            off += 8;
            return v;
        };
        Header h;
        const usize ml = le64();
        if (off + ml > n)
            throw Panic(HF_ERR_BAD_ARG, "header: truncated magic");
        h.magic.assign(p + off, p + off + ml);
        off += ml;
        const usize cnt = le64();
        if (cnt > (n - off) / 10) // an entry is 10 bytes: a count beyond the buffer is garbage
            throw Panic(HF_ERR_BAD_ARG, "header: table length exceeds the header");
        for (usize i = 0; i < cnt; ++i) {
            const usize key = le64();
            if (off + 2 > n)
                throw Panic(HF_ERR_BAD_ARG, "header: truncated table");
            h.btree[key] = {p[off], p[off + 1]};
            off += 2;
        }
        h.sentinel = le64();
        h.readbytes = le64();
        return h;
    }
    static Header from_binary(const std::vector<u8> &v) { return from_binary(v.data(), v.size()); }
    void update_readbytes(std::uint64_t bytes) { readbytes = bytes; } // header.rs:47-49
    bool operator==(const Header &o) const
    {
        return magic == o.magic && btree == o.btree && sentinel == o.sentinel &&
               readbytes == o.readbytes;
    }
};

} // namespace header

// ------------------------------------------------ src/huffman/decode/mod.rs

namespace decode {

// prepare_lookup (decode/mod.rs:140-155): the values in key order plus the
// sorted keys (the reference keeps them as a rank dictionary over a bit
// vector; `rank1(v + 1) - 1` is the index of the largest key <= v).
inline std::pair<std::vector<std::pair<u8, u8>>, std::vector<usize>>
prepare_lookup(const std::map<usize, std::pair<u8, u8>> &bt)
{
    std::vector<std::pair<u8, u8>> table;
    std::vector<usize> keys;
    for (auto &kv : bt) {
        keys.push_back(kv.first);
        table.push_back(kv.second);
    }
    return {table, keys};
}

// search_key_or_next_small_key (decode/mod.rs:157-165)
inline std::pair<u8, u8> search_key_or_next_small_key(const std::map<usize, std::pair<u8, u8>> &tree,
                                                      usize key)
{
    auto it = tree.upper_bound(key);
    if (it == tree.begin())
        throw Panic(HF_ERR_DECODE, "Panic!!!!");
    return std::prev(it)->second;
}

// read (decode/mod.rs:167-231): slice decoder.  Without a gap array the walk
// is sequential on the device; pass the encoder's gap array for the
// chunk-parallel kernel.
inline std::vector<u8> read(const u8 *data, std::size_t n, const Model &model, usize goalsbyte,
                            const GapArray *gap = nullptr)
{
    DeviceModel dm(model);
    std::vector<u8> out(goalsbyte);
    const bool use_gap = gap && goalsbyte > 0 && gap->fits(goalsbyte, n);
    Runtime::check(hf_decode_host(Runtime::ctx(), dm.handle(), data, n, goalsbyte,
                                  use_gap ? gap->chunk_syms : 64,
                                  use_gap ? gap->chunk_bits.data() : nullptr,
                                  use_gap ? gap->seg_off.data() : nullptr, out.data()),
                   "decode::read");
    return out;
}
inline std::vector<u8> read(const std::vector<u8> &data, const Model &model, usize goalsbyte,
                            const GapArray *gap = nullptr)
{
    return read(data.data(), data.size(), model, goalsbyte, gap);
}

// struct Decoder<R: Read> (decode/mod.rs:15-26): streaming decoder.  read()
// hands out decoded bytes piece by piece: the inner reader is drained only as
// far as the next piece needs, the GPU decodes that piece, and any read size
// (the reference's example uses 15 bytes) is a memcpy out of it.  The host
// holds a window of the stream and one piece of output, whatever the stream's
// size.  With the encoder's gap array (checked with GapArray::fits first) a
// piece is a run of whole segments; without one -- a file the reference wrote
// -- the device finds the code boundaries of each window itself.
template <class R>
class Decoder {
  public:
    // with a gap array: symbols per piece; without: stream bytes per window
    static constexpr usize PIECE_SYMS = (usize(64) << 20) / HFGPU_PIECE_DIVISOR;
    static constexpr std::size_t WINDOW_BYTES = (std::size_t(32) << 20) / HFGPU_PIECE_DIVISOR;

    // Decoder::new (decode/mod.rs:29-47)
    template <class W>
    Decoder(R reader, encode::Encoder<W> &encoder)
        : inner_(std::move(reader)), goalsbyte_(encoder.readbytes),
          dm_(std::make_unique<DeviceModel>(*encoder.model)), gap_(encoder.gap())
    {
        use_gap_ = goalsbyte_ > 0 && gap_.fits(goalsbyte_, std::nullopt);
    }
    // Decoder::from_header (decode/mod.rs:48-66); stream_bytes = length of the
    // stream behind the reader when the caller knows it (a file)
    Decoder(const header::Header &header, R reader, const GapArray *gap = nullptr,
            std::optional<usize> stream_bytes = std::nullopt)
        : inner_(std::move(reader)), goalsbyte_(header.readbytes),
          dm_(std::make_unique<DeviceModel>(header.btree, header.sentinel))
    {
        if (gap && goalsbyte_ > 0 && gap->fits(goalsbyte_, stream_bytes)) {
            gap_ = *gap;
            use_gap_ = true;
        }
    }
    bool uses_gap_array() const { return use_gap_; }
    // impl Read::read (decode/mod.rs:70-137)
    std::size_t read(u8 *buf, std::size_t len)
    {
        std::size_t done = 0;
        while (done < len) {
            if (pos_ == out_.size() && !next_piece())
                break;
            const std::size_t k = std::min<std::size_t>(len - done, out_.size() - pos_);
            std::memcpy(buf + done, out_.data() + pos_, k);
            pos_ += k;
            done += k;
        }
        return done;
    }
    std::size_t read_to_end(std::vector<u8> &dst)
    {
        std::size_t total = 0;
        while (pos_ < out_.size() || next_piece()) {
            dst.insert(dst.end(), out_.begin() + (std::ptrdiff_t)pos_, out_.end());
            total += out_.size() - pos_;
            pos_ = out_.size();
        }
        return total;
    }

  private:
    // Reads from the inner reader until the window reaches stream byte `upto`
    // (exclusive) or the reader ends.
    void fill_to(usize upto)
    {
        while (!eof_ && comp_base_ + comp_.size() < upto) {
            const std::size_t have = comp_.size();
            const std::size_t want = (std::size_t)std::min<usize>(upto - comp_base_ - have, 1u << 22);
            comp_.resize(have + want);
            const std::size_t n = inner_.read(comp_.data() + have, want);
            comp_.resize(have + n);
            if (n == 0)
                eof_ = true;
        }
    }
    void drop_before(usize byte)
    {
        if (byte > comp_base_) {
            const std::size_t k = (std::size_t)std::min<usize>(byte - comp_base_, comp_.size());
            comp_.erase(comp_.begin(), comp_.begin() + (std::ptrdiff_t)k);
            comp_base_ += k;
        }
    }
    bool next_piece()
    {
        out_.clear();
        pos_ = 0;
        if (next_sym_ >= goalsbyte_)
            return false;
        return use_gap_ ? piece_with_gap() : piece_from_window();
    }
    bool piece_with_gap()
    {
        const usize seg_syms = (usize)HF_SEG_CHUNKS * gap_.chunk_syms;
        const usize s0 = next_sym_ / seg_syms, n_segs = gap_.seg_off.size();
        const usize s1 = std::min<usize>(n_segs, s0 + std::max<usize>(1, PIECE_SYMS / seg_syms));
        const usize n_piece = std::min(goalsbyte_, s1 * seg_syms) - next_sym_;
        const usize end_bit = s1 < n_segs ? gap_.seg_off[s1] : gap_.total_bits();
        const usize first_byte = (gap_.seg_off[s0] / 8) & ~usize(15);
        drop_before(first_byte);
        fill_to((end_bit + 7) / 8 + 16);
        if (comp_base_ != first_byte || comp_base_ + comp_.size() < (end_bit + 7) / 8)
            throw Panic(HF_ERR_DECODE, "Decoder::read: the stream ends before the gap array does");
        out_.resize(n_piece);
        Runtime::check(hf_decode_range_host(Runtime::ctx(), dm_->handle(), comp_.data(), comp_.size(),
                                            first_byte * 8, n_piece, gap_.chunk_syms,
                                            gap_.chunk_bits.data() + s0 * HF_SEG_CHUNKS,
                                            gap_.seg_off.data() + s0, out_.data()),
                       "Decoder::read");
        next_sym_ += n_piece;
        return true;
    }
    bool piece_from_window()
    {
        std::size_t window = WINDOW_BYTES;
        while (true) {
            fill_to(comp_base_ + window);
            const bool last = eof_;
            // a code has at least one bit: the window cannot hold more symbols than bits
            const usize n_max = std::min<usize>({goalsbyte_ - next_sym_, (usize)comp_.size() * 8 + 8,
                                                 usize(256) << 20});
            const bool all = last && n_max == goalsbyte_ - next_sym_;
            out_.resize(n_max);
            usize n = 0, end_bit = 0;
            Runtime::check(hf_decode_window_host(Runtime::ctx(), dm_->handle(), comp_.data(),
                                                 comp_.size(), start_bit_, n_max, all ? 1 : 0,
                                                 out_.data(), &n, &end_bit),
                           "Decoder::read");
            if (n == 0 && !all) {
                if (last) // the window is the rest of the stream and more symbols than fit a piece remain
                    throw Panic(HF_ERR_DECODE, "Decoder::read: window without a whole segment");
                window *= 2; // very long codes: not one whole segment inside the window yet
                continue;
            }
            out_.resize(n);
            next_sym_ += n;
            start_bit_ = (std::uint32_t)(end_bit % 8);
            drop_before(comp_base_ + end_bit / 8);
            return n != 0;
        }
    }

    R inner_;
    usize goalsbyte_;
    std::unique_ptr<DeviceModel> dm_;
    GapArray gap_;
    bool use_gap_ = false;
    std::vector<u8> comp_; // window of the stream, from stream byte comp_base_
    usize comp_base_ = 0;
    bool eof_ = false;
    usize next_sym_ = 0;          // first symbol of the next piece
    std::uint32_t start_bit_ = 0; // without a gap array: bit of comp_[0] at which it starts
    std::vector<u8> out_;         // the current piece
    std::size_t pos_ = 0;
};

// vault::Decoder (decode/vault.rs:28-39,65-86): the reference's experimental
// third decoder.  Only its results are reproduced (same bytes as Decoder);
// its precondition `sentinel <= 8` (vault.rs:52-56) is kept.
namespace vault {
template <class R>
class Decoder {
  public:
    Decoder(R reader, const Model &model, std::uint64_t output)
        : inner_(std::move(reader)), remaining_(output)
    {
        if (model.sentinel() > 8)
            throw Panic(HF_ERR_UNSUPPORTED, "assertion failed: sentinel <= 8");
        header::Header h;
        h.btree = model.to_btreemap();
        h.sentinel = model.sentinel();
        h.readbytes = output;
        dec_ = std::make_unique<decode::Decoder<R>>(h, std::move(inner_));
    }
    std::size_t read(u8 *buf, std::size_t len) { return dec_->read(buf, len); }
    std::size_t read_to_end(std::vector<u8> &dst) { return dec_->read_to_end(dst); }

  private:
    R inner_;
    std::uint64_t remaining_;
    std::unique_ptr<decode::Decoder<R>> dec_;
};
} // namespace vault

} // namespace decode
} // namespace huffman

// --------------------------------------------------------------- adaptors

// Cursor<Vec<u8>> as a Writer
struct VecWriter {
    std::vector<u8> buf;
    std::size_t write(const u8 *p, std::size_t n)
    {
        buf.insert(buf.end(), p, p + n);
        return n;
    }
    void flush() {}
    const std::vector<u8> &get_ref() const { return buf; }
};
// Cursor<&[u8]> as a (seekable) Reader
struct SliceReader {
    const u8 *p = nullptr;
    std::size_t n = 0, pos = 0;
    SliceReader(const u8 *d, std::size_t len) : p(d), n(len) {}
    explicit SliceReader(const std::vector<u8> &v) : p(v.data()), n(v.size()) {}
    std::size_t read(u8 *dst, std::size_t len)
    {
        const std::size_t k = std::min(len, n - pos);
        std::memcpy(dst, p + pos, k);
        pos += k;
        return k;
    }
    void rewind() { pos = 0; }
};
struct FileReader {
    std::ifstream f;
    explicit FileReader(const std::string &path) : f(path, std::ios::binary)
    {
        if (!f)
            throw Panic(HF_ERR_BAD_ARG, "Failed to open source file");
    }
    std::size_t read(u8 *dst, std::size_t len)
    {
        f.read(reinterpret_cast<char *>(dst), (std::streamsize)len);
        return (std::size_t)f.gcount();
    }
    void rewind()
    {
        f.clear();
        f.seekg(0);
    }
};
struct FileWriter {
    std::ofstream f;
    explicit FileWriter(const std::string &path) : f(path, std::ios::binary | std::ios::trunc)
    {
        if (!f)
            throw Panic(HF_ERR_BAD_ARG, "Failed to create destination file");
    }
    std::size_t write(const u8 *p, std::size_t n)
    {
        f.write(reinterpret_cast<const char *>(p), (std::streamsize)n);
        return n;
    }
    void flush() { f.flush(); }
};

// ------------------------------------------------------------- src/lib.rs

// u64_to_bytes / bytes_to_u64 (src/lib.rs:82-102): big endian
inline std::array<u8, 8> u64_to_bytes(std::uint64_t num)
{
    std::array<u8, 8> b{};
    for (int i = 0; i < 8; ++i)
        b[i] = (u8)(num >> (56 - 8 * i));
    return b;
}
inline std::uint64_t bytes_to_u64(const u8 *bytes)
{
    std::uint64_t r = 0;
    for (int i = 0; i < 8; ++i)
        r += (std::uint64_t)bytes[i] << (56 - i * 8);
    return r;
}

// Side file that carries the gap array of <destination>: "hfgap1\0\0",
// u32 chunk_syms, u64 symbols, u64 n_chunks, u64 n_segs, chunk_bits, seg_off
// (little endian).  The compressed file itself stays byte-identical to what
// the reference writes.
inline std::string gap_sidecar_path(const std::string &path) { return path + ".hfgap"; }
inline void write_gap_sidecar(const std::string &path, const huffman::GapArray &g)
{
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    const char tag[8] = {'h', 'f', 'g', 'a', 'p', '1', 0, 0};
    const std::uint64_t nc = g.chunk_bits.size(), ns = g.seg_off.size();
    f.write(tag, 8);
    f.write(reinterpret_cast<const char *>(&g.chunk_syms), 4);
    f.write(reinterpret_cast<const char *>(&g.symbols), 8);
    f.write(reinterpret_cast<const char *>(&nc), 8);
    f.write(reinterpret_cast<const char *>(&ns), 8);
    f.write(reinterpret_cast<const char *>(g.chunk_bits.data()), (std::streamsize)(nc * 2));
    f.write(reinterpret_cast<const char *>(g.seg_off.data()), (std::streamsize)(ns * 8));
}
inline bool read_gap_sidecar(const std::string &path, huffman::GapArray &g)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f)
        return false;
    const std::uint64_t size = (std::uint64_t)f.tellg();
    f.seekg(0);
    char tag[8];
    std::uint64_t nc = 0, ns = 0;
    if (size < 36 || !f.read(tag, 8) || std::memcmp(tag, "hfgap1", 6) != 0)
        return false;
    f.read(reinterpret_cast<char *>(&g.chunk_syms), 4);
    f.read(reinterpret_cast<char *>(&g.symbols), 8);
    f.read(reinterpret_cast<char *>(&nc), 8);
    f.read(reinterpret_cast<char *>(&ns), 8);
    // the counts must account for the file exactly: nothing is allocated for a
    // truncated or foreign file
    if (!f || nc > size / 2 || ns > size / 8 || 36 + nc * 2 + ns * 8 != size)
        return false;
    g.chunk_bits.resize(nc);
    g.seg_off.resize(ns);
    f.read(reinterpret_cast<char *>(g.chunk_bits.data()), (std::streamsize)(nc * 2));
    f.read(reinterpret_cast<char *>(g.seg_off.data()), (std::streamsize)(ns * 8));
    return (bool)f;
}

// stream_compress_with_header_information (src/lib.rs:31-80):
// "pzhf" || u64-BE header length || bincode(Header) || bitstream.
inline void stream_compress_with_header_information(const std::string &source,
                                                    const std::string &destination)
{
    FileReader reader(source);
    {
        // A file that fits one staging buffer goes to the device once: counting
        // pass, codebook, planned encode and download in a single call.  The
        // bytes written are the same as on the general path below.
        std::ifstream sz(source, std::ios::binary | std::ios::ate);
        const std::uint64_t size = sz ? (std::uint64_t)sz.tellg() : 0;
        constexpr std::uint64_t SINGLE_UPLOAD_LIMIT = (std::uint64_t(1) << 30) / HFGPU_PIECE_DIVISOR;
        if (size > 0 && size <= SINGLE_UPLOAD_LIMIT) {
            std::vector<u8> data(size);
            if (reader.read(data.data(), size) != size)
                throw IoError(false, "failed to read the source file");
            std::array<usize, 256> codes{}, lens{};
            std::vector<u8> stream(size + 16);
            huffman::GapArray gap;
            gap.chunk_syms = 64;
            gap.symbols = size;
            std::uint64_t nc = 0, ns = 0;
            Runtime::check(hf_gap_layout(size, gap.chunk_syms, &nc, &ns), "gap layout");
            gap.chunk_bits.resize(nc);
            gap.seg_off.resize(ns);
            std::uint64_t bits = 0;
            Runtime::check(hf_compress_host(Runtime::ctx(), data.data(), size, codes.data(),
                                            lens.data(), stream.data(), stream.size(),
                                            gap.chunk_syms, gap.chunk_bits.data(),
                                            gap.seg_off.data(), &bits),
                           "stream_compress");
            const huffman::Huffman h(codes, lens);
            huffman::header::Header hd;
            hd.magic = {'p', 'z', 'h', 'f'}; // Encoder::magic, encode.rs:51-53
            hd.btree = h.to_btreemap();
            hd.sentinel = h.sentinel();
            hd.update_readbytes(size); // lib.rs:52-53
            const auto header = hd.to_binary();
            const auto header_length = u64_to_bytes(header.size());
            FileWriter out(destination);
            out.write(hd.magic.data(), hd.magic.size());
            out.write(header_length.data(), 8);
            out.write(header.data(), header.size());
            out.write(stream.data(), (bits + 7) / 8);
            out.flush();
            write_gap_sidecar(gap_sidecar_path(destination), gap);
            return;
        }
    }
    std::vector<u8> buffer(BUF);
    auto h = huffman::Huffman::from_reader(reader);
    huffman::encode::Encoder<FileWriter> writer(FileWriter(destination), h);
    writer.track_write_value = false; // the return value of write() is unused here
    auto hd = huffman::header::Header::from(writer);
    std::ifstream sz(source, std::ios::binary | std::ios::ate);
    hd.update_readbytes((std::uint64_t)sz.tellg()); // lib.rs:52-53
    const auto header = hd.to_binary();
    const auto header_length = u64_to_bytes(header.size());
    writer.plain_write(writer.magic());
    writer.plain_write(header_length.data(), 8);
    writer.plain_write(header);
    while (true) { // lib.rs:68-77
        const std::size_t n = reader.read(buffer.data(), buffer.size());
        if (n == 0)
            break;
        writer.write(buffer.data(), n);
    }
    writer.flush();
    write_gap_sidecar(gap_sidecar_path(destination), writer.gap());
}

// stream_decompress_with_header_information (src/lib.rs:28-30) is
// `unimplemented!()` in the reference; this is its natural completion through
// Header::from_binary + Decoder::from_header.  Uses the gap side file when it
// sits next to the source, else the sequential device walk.
inline void stream_decompress_with_header_information(const std::string &source,
                                                      const std::string &destination)
{
    std::uint64_t file_size = 0;
    {
        std::ifstream sz(source, std::ios::binary | std::ios::ate);
        if (sz)
            file_size = (std::uint64_t)sz.tellg();
    }
    FileReader reader(source);
    u8 fixed[12];
    if (reader.read(fixed, 12) != 12 || std::memcmp(fixed, "pzhf", 4) != 0)
        throw Panic(HF_ERR_BAD_ARG, "not a pzhf file");
    const std::uint64_t hlen = bytes_to_u64(fixed + 4);
    if (hlen > file_size - 12) // (nothing is allocated for a length the file cannot hold)
        throw Panic(HF_ERR_BAD_ARG, "header length exceeds the file");
    std::vector<u8> hb(hlen);
    if (reader.read(hb.data(), hlen) != hlen)
        throw Panic(HF_ERR_BAD_ARG, "truncated header");
    const auto hd = huffman::header::Header::from_binary(hb);
    huffman::GapArray gap;
    const bool have_gap = read_gap_sidecar(gap_sidecar_path(source), gap);
    // the side file is used only if it fits this very stream (GapArray::fits)
    huffman::decode::Decoder<FileReader> dec(hd, std::move(reader), have_gap ? &gap : nullptr,
                                             file_size - 12 - hlen);
    FileWriter out(destination);
    std::vector<u8> piece(std::size_t(16) << 20);
    while (true) {
        const std::size_t n = dec.read(piece.data(), piece.size());
        if (n == 0)
            break;
        out.write(piece.data(), n);
    }
    out.flush();
}

} // namespace rscompress_huffman
#endif // HFGPU_HPP
