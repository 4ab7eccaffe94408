// hfgpu: HIP kernels (gfx950 / CDNA4, wave64) and the C ABI around them.  The
// kernels live in kern_hist.hpp, kern_encode.hpp, kern_decode.hpp and
// kern_gap.hpp (one translation unit); this file holds the host side.
//
// Everything here is HBM-bound byte/bit work: no matrix cores.  The design
// rules are coalesced 16-byte global accesses, conflict-free LDS layouts, one
// pass over the input where the data flow allows it, and enough workgroups in
// flight (>> 256 CUs) to cover HBM latency.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include "hfgpu.h"

#include "hfgpu_internal.h"

#include "dev_common.hpp"
#include "kern_hist.hpp"
#include "kern_encode.hpp"
#include "kern_decode.hpp"
#include "kern_gap.hpp"

// Scratch of hf_encode_noalloc_async: a fixed block in the code object's own
// device memory, so that entry point never calls the allocator.  Every GPU the
// library runs on holds its own copy at its own address for the life of the
// process.  In host code the array's name is only a handle for
// hipGetSymbolAddress: hf_ctx_create resolves the address on the context's
// device and everything else goes through that pointer.
static __device__ __align__(256) unsigned char scratch[HF_NOALLOC_SCRATCH_BYTES];

namespace {

// One entry per GPU, shared by every context of the process on that GPU: the
// block's device address and the event its last user recorded behind its last
// kernel.  A call on another stream waits for that event before its first
// kernel (neither the wait nor the record allocates); calls on one stream are
// ordered anyway.  Made by the first hf_ctx_create on the device.
struct FixedScratch {
    void *ptr = nullptr;
    hipEvent_t done = nullptr;
    hipStream_t last_stream = nullptr;
    bool used = false;
};
constexpr int FIXED_SCRATCH_DEVICES = 64;
FixedScratch g_fixed_scratch[FIXED_SCRATCH_DEVICES];
std::mutex g_fixed_scratch_mutex;

} // namespace

// ===================================================================== host

struct hf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // The scratch block is shared by every call on this context, whatever stream
    // the caller has bound: `scratch_done` is recorded behind the last kernel
    // that uses it, and a call on another stream waits for that event first.
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_busy = false;
    uint64_t *d_result = nullptr; // result block of the synchronous entry points
    uint64_t *h_small = nullptr;  // 8 pinned words: where small results land (a copy into pageable
                                  // memory goes through the runtime's staging buffer and costs tens of microseconds)
    void *fixed_scratch = nullptr; // this device's copy of `scratch` (hipGetSymbolAddress)
    int num_cus = 256;
    int enc_mode = 1;
    int hist_grid = 0;
    int timeout_ms = 2000;
    int debug = 0;
    uint64_t *dec_trace = nullptr; // device buffer for the decode kernel's phase stamps
    // device blocks of destroyed models, kept for the next hf_model_create: a
    // codec that builds a model per stream would otherwise pay hipMalloc/hipFree each time
    struct Blob {
        void *d;
        size_t bytes;
        hipEvent_t free_after; // recorded behind the last kernels of the model that owned it
                               // (null: they have been waited for)
    };
    std::vector<Blob> blob_cache;
    // pinned staging of the table uploads, used in turn: the upload is an
    // asynchronous copy on the context's stream, `done` tells when a buffer
    // may be filled again
    struct Pinned {
        void *h = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool used = false;
    };
    Pinned pinned[4];
    unsigned pinned_next = 0;
    void *arena = nullptr; // device staging of the *_host entry points (grow only)
    size_t arena_bytes = 0;
    // hf_decode_host moves large buffers in pieces: upload, decode and download
    // of consecutive pieces overlap on three streams
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    int host_piece_mib = 64;
    // option "kernel_timing": HIP events around the dominant kernel of every
    // histogram / encode / decode call (on the stream it is launched on), kept
    // in a ring until hf_ctx_kernel_ms collects them
    struct Timed {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ring;
        size_t pending = 0; // pairs recorded since the last collection
    };
    Timed timed[HF_TIMED_KINDS];
    bool kernel_timing = false;
    const char *encode_kernel = ""; // what the last encode launched over its complete batches
    const char *decode_kernel = ""; // what the last hf_decode[_async] launched
};

struct hf_model {
    hf_ctx *ctx = nullptr;
    void *d_blob = nullptr;
    size_t blob_bytes = 0;
    uint64_t *d_enc = nullptr;   // 256 x (code << 6 | len)
    uint2 *d_enc2 = nullptr;     // 256 x {code, len} ({0, ENC_LONG_FLAG | len} above 32 bits)
    uint8_t *d_len8 = nullptr;   // 256 x len
    uint16_t *d_lut = nullptr;
    uint32_t *d_pair = nullptr;  // two-symbol table (only when pair_entries != 0)
    uint32_t pair_entries = 0;
    uint64_t *d_keys = nullptr;
    uint8_t *d_ksyms = nullptr;
    uint8_t *d_klens = nullptr;
    uint32_t lut_entries = 0;
    uint32_t m = 0;
    uint32_t sentinel = 0;
    uint32_t w1 = 0;
    uint32_t maxlen = 0;
    int enc_group = 1; // codes the encoder merges per append: 4, 2 or 1
    bool enc_spec = false; // a merged group can exceed 32 bits
    bool has_encode = false;
    bool bytewise = false;    // every one of the 256 symbols has an 8-bit code: encoding is a byte map
    bool lut_escapes = false; // the decode table holds entries with the escape bit
    double mean_bits = 8.0; // code length of a symbol drawn with probability 2^-len
    bool regular_chunks = false; // 64-symbol chunks sit on few LDS banks unless the image is padded
    uint32_t dec_spec_k = 0; // 16, 8, 7 or 6: codes per 64-bit window the decoder may assume (else 0)
    hipStream_t use_stream = nullptr; // the tables were uploaded on this stream ...
    hipEvent_t ready = nullptr;       // ... and are complete once this event has passed
    mutable bool foreign = false;     // kernels on other streams have used the tables
};

namespace {

int fail(hf_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return status;
}

#define HF_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((ctx), HF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

// Makes the thread's current device the context's for the lifetime of the
// guard: every entry point that reaches the HIP runtime starts with one, so a
// process that holds contexts for several GPUs allocates and launches on the
// right one and finds its own current device unchanged afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const hf_ctx *ctx)
    {
        if (ctx && hipGetDevice(&prev) == hipSuccess && prev != ctx->device)
            switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// The context's scratch block with at least `bytes` bytes, ordered behind its
// previous user: work queued on another stream may still be reading or writing
// it, so the context's current stream waits for the event that user left (the
// same stream is ordered already), and the block is only freed for a larger
// one once that event has passed.  Every user calls scratch_release() after
// its last kernel that touches the block.
int scratch_acquire(hf_ctx *ctx, size_t bytes)
{
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            if (ctx->scratch_busy)
                HF_HIP(ctx, hipEventSynchronize(ctx->scratch_done));
            HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            HF_HIP(ctx, hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
            ctx->scratch_busy = false;
        }
        size_t want = std::max(bytes, (size_t)1 << 20);
        HF_HIP(ctx, hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    if (ctx->scratch_busy && ctx->scratch_stream != ctx->stream)
        HF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->scratch_done, 0));
    return HF_OK;
}

int scratch_release(hf_ctx *ctx)
{
    if (!ctx->scratch_done)
        HF_HIP(ctx, hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming));
    HF_HIP(ctx, hipEventRecord(ctx->scratch_done, ctx->stream));
    ctx->scratch_stream = ctx->stream;
    ctx->scratch_busy = true;
    return HF_OK;
}

int ensure_arena(hf_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->arena_bytes)
        return HF_OK;
    if (ctx->arena) {
        HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        HF_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
    }
    const size_t want = bytes + bytes / 4 + 4096;
    HF_HIP(ctx, hipMalloc(&ctx->arena, want));
    ctx->arena_bytes = want;
    return HF_OK;
}

int flags_to_status(hf_ctx *ctx, uint64_t flags)
{
    if (flags & HF_FLAG_TIMEOUT)
        return fail(ctx, HF_ERR_INTERNAL, "a device-side inter-workgroup wait expired");
    if (flags & HF_FLAG_PLAN)
        return fail(ctx, HF_ERR_BAD_ARG, "the batch histograms do not belong to this input");
    if (flags & HF_FLAG_CAPACITY)
        return fail(ctx, HF_ERR_CAPACITY, "output buffer too small");
    if (flags & HF_FLAG_DECODE)
        return fail(ctx, HF_ERR_DECODE, "bit window below the smallest key of the model");
    return HF_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int upload_model(hf_ctx *ctx, hf_model *mod, const uint64_t *enc_tab /*256 or null*/,
                 const uint32_t *enc_tab2 /*512*/, const hf::DecodeLut &lut, const uint64_t *keys,
                 const uint8_t *syms, const uint8_t *klens, uint64_t m)
{
    const size_t off_enc = 0;
    const size_t off_enc2 = off_enc + 256 * sizeof(uint64_t);
    const size_t off_len8 = off_enc2 + 512 * sizeof(uint32_t);
    const size_t off_lut = align_up(off_len8 + 256, 16);
    const size_t off_keys = align_up(off_lut + lut.pool.size() * 2, 16);
    const size_t off_ksyms = off_keys + 256 * sizeof(uint64_t);
    const size_t off_klens = off_ksyms + 256;
    // two-symbol decode table: only for complete prefix codes in which nearly every
    // first-level index holds two complete codes (strongly skewed streams)
    std::vector<uint32_t> pair;
    const bool complete = hf::is_complete_prefix_code(keys, klens, m, lut.sentinel);
    if (lut.w1 >= 6 && complete)
        (void)hf::build_pair_lut(lut, 0.01, keys, syms, klens, m,
                                 (ctx->debug & 2048) ? 0u : HF_PAIR_MAX_SUB, &pair);
    const size_t off_pair = align_up(off_klens + 256, 16);
    const size_t total = off_pair + pair.size() * sizeof(uint32_t);
    hf_ctx::Pinned &ps = ctx->pinned[ctx->pinned_next++ % 4];
    if (ps.used)
        HF_HIP(ctx, hipEventSynchronize(ps.done)); // the copy that last read this buffer is over
    if (ps.bytes < total) {
        if (ps.h)
            HF_HIP(ctx, hipHostFree(ps.h));
        ps.h = nullptr, ps.bytes = 0;
        HF_HIP(ctx, hipHostMalloc(&ps.h, align_up(total, 4096), hipHostMallocDefault));
        ps.bytes = align_up(total, 4096);
    }
    if (!ps.done)
        HF_HIP(ctx, hipEventCreateWithFlags(&ps.done, hipEventDisableTiming));
    uint8_t *blob = static_cast<uint8_t *>(ps.h);
    std::memset(blob, 0, total);
    if (enc_tab) {
        std::memcpy(blob + off_enc, enc_tab, 256 * sizeof(uint64_t));
        std::memcpy(blob + off_enc2, enc_tab2, 512 * sizeof(uint32_t));
        for (int i = 0; i < 256; ++i)
            blob[off_len8 + i] = (uint8_t)(enc_tab[i] & 63u);
    }
    std::memcpy(blob + off_lut, lut.pool.data(), lut.pool.size() * 2);
    std::memcpy(blob + off_keys, keys, m * sizeof(uint64_t));
    std::memcpy(blob + off_ksyms, syms, m);
    std::memcpy(blob + off_klens, klens, m);
    if (!pair.empty())
        std::memcpy(blob + off_pair, pair.data(), pair.size() * sizeof(uint32_t));
    for (size_t i = 0; i < ctx->blob_cache.size(); ++i) {
        if (ctx->blob_cache[i].bytes >= total) {
            // kernels of the block's previous owner may still be running: the copy
            // below is ordered behind the event that owner left (no stream handle is
            // kept, the stream may be gone by now)
            if (ctx->blob_cache[i].free_after) {
                HF_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->blob_cache[i].free_after, 0));
                (void)hipEventDestroy(ctx->blob_cache[i].free_after);
                ctx->blob_cache[i].free_after = nullptr;
            }
            mod->d_blob = ctx->blob_cache[i].d;
            mod->blob_bytes = ctx->blob_cache[i].bytes;
            ctx->blob_cache.erase(ctx->blob_cache.begin() + (long)i);
            break;
        }
    }
    if (!mod->d_blob) {
        mod->blob_bytes = align_up(total, 4096);
        HF_HIP(ctx, hipMalloc(&mod->d_blob, mod->blob_bytes));
    }
    HF_HIP(ctx, hipMemcpyAsync(mod->d_blob, blob, total, hipMemcpyHostToDevice, ctx->stream));
    HF_HIP(ctx, hipEventRecord(ps.done, ctx->stream));
    ps.used = true;
    mod->use_stream = ctx->stream;
    mod->ready = ps.done;
    uint8_t *b = static_cast<uint8_t *>(mod->d_blob);
    mod->d_enc = reinterpret_cast<uint64_t *>(b + off_enc);
    mod->d_enc2 = reinterpret_cast<uint2 *>(b + off_enc2);
    mod->d_len8 = b + off_len8;
    mod->d_lut = reinterpret_cast<uint16_t *>(b + off_lut);
    mod->d_keys = reinterpret_cast<uint64_t *>(b + off_keys);
    mod->d_ksyms = b + off_ksyms;
    mod->d_klens = b + off_klens;
    mod->d_pair = pair.empty() ? nullptr : reinterpret_cast<uint32_t *>(b + off_pair);
    mod->pair_entries = (uint32_t)pair.size();
    mod->lut_entries = (uint32_t)lut.pool.size();
    mod->m = (uint32_t)m;
    mod->sentinel = lut.sentinel;
    mod->w1 = lut.w1;
    mod->lut_escapes = lut.has_escape;
    // spread of a 64-symbol chunk's bit length when symbols occur with the
    // probabilities 2^-len that the lengths imply
    double kraft = 0, m1 = 0, m2 = 0;
    for (uint64_t i = 0; i < m; ++i) {
        const double w = std::ldexp(1.0, -(int)klens[i]);
        kraft += w, m1 += w * klens[i], m2 += w * klens[i] * klens[i];
    }
    if (kraft > 0) {
        m1 /= kraft, m2 /= kraft;
        mod->mean_bits = m1;
        mod->regular_chunks = hf::prefer_padded_image(64.0 * m1, std::sqrt(64.0 * std::max(0.0, m2 - m1 * m1)));
    }
    mod->dec_spec_k = hf::pick_decode_window(keys, klens, m, lut.sentinel);
    return HF_OK;
}

bool valid_chunk_syms(uint32_t cs)
{
    return cs == 64 || cs == 128 || cs == 256;
}

// Kernel timing (option "kernel_timing"): the next free pair of the ring, or
// nothing when the option is off or the ring is full.
constexpr size_t TIMED_RING = 4096;
std::pair<hipEvent_t, hipEvent_t> *timed_slot(hf_ctx *ctx, int kind)
{
    if (!ctx->kernel_timing)
        return nullptr;
    hf_ctx::Timed &t = ctx->timed[kind];
    if (t.pending >= TIMED_RING)
        return nullptr;
    if (t.pending >= t.ring.size()) {
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess)
            return nullptr;
        t.ring.push_back(ev);
    }
    return &t.ring[t.pending++];
}
#define HF_TIMED_LAUNCH(ctx, kind, launch)                                                    \
    do {                                                                                      \
        std::pair<hipEvent_t, hipEvent_t> *ev_ = timed_slot(ctx, kind);                       \
        if (ev_)                                                                              \
            (void)hipEventRecord(ev_->first, (ctx)->stream);                                  \
        launch;                                                                               \
        if (ev_)                                                                              \
            (void)hipEventRecord(ev_->second, (ctx)->stream);                                 \
    } while (0)

// Work that reads a model's tables must come after their upload: on the
// stream that carried the upload that is the case already, any other stream
// waits for the upload's event (and the model remembers the foreign use).
int model_on_stream(hf_ctx *ctx, const hf_model *model)
{
    if (model->ready && ctx->stream != model->use_stream) {
        HF_HIP(ctx, hipStreamWaitEvent(ctx->stream, model->ready, 0));
        model->foreign = true;
    }
    return HF_OK;
}

} // namespace

extern "C" {

int hf_version(void) { return 1; }

const char *hf_strerror(int status)
{
    switch (status) {
    case HF_OK: return "ok";
    case HF_ERR_BAD_ARG: return "bad argument";
    case HF_ERR_CODELEN: return "code length above 64 (InvalidData)";
    case HF_ERR_EMPTY: return "empty input or model";
    case HF_ERR_CAPACITY: return "output buffer too small";
    case HF_ERR_HIP: return "HIP runtime error";
    case HF_ERR_BAD_CODEBOOK: return "code does not fit its length";
    case HF_ERR_DECODE: return "bit window below the smallest key";
    case HF_ERR_INTERNAL: return "device-side wait expired";
    case HF_ERR_UNSUPPORTED: return "code length or sentinel above 57";
    default: return "unknown status";
    }
}

int hf_ctx_create(int device, hf_ctx **out)
{
    if (!out)
        return HF_ERR_BAD_ARG;
    *out = nullptr;
    hf_ctx *ctx = new (std::nothrow) hf_ctx();
    if (!ctx)
        return HF_ERR_INTERNAL;
    ctx->device = device;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        delete ctx;
        return HF_ERR_HIP;
    }
    DeviceGuard on(ctx); // the caller's current device is left as it was
    int cur = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e != hipSuccess || cur != device) {
        delete ctx;
        return HF_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        ctx->num_cus = prop.multiProcessorCount;
    e = hipMalloc(reinterpret_cast<void **>(&ctx->d_result), HF_RES_WORDS * sizeof(uint64_t));
    if (e != hipSuccess) {
        delete ctx;
        return HF_ERR_HIP;
    }
    e = hipHostMalloc(reinterpret_cast<void **>(&ctx->h_small), 8 * sizeof(uint64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipFree(ctx->d_result);
        delete ctx;
        return HF_ERR_HIP;
    }
    // this device's address of the fixed scratch block (the device is current here)
    // and the event that orders its users
    if (device < FIXED_SCRATCH_DEVICES) {
        std::lock_guard<std::mutex> lock(g_fixed_scratch_mutex);
        FixedScratch &fs = g_fixed_scratch[device];
        e = hipSuccess;
        if (!fs.ptr)
            e = hipGetSymbolAddress(&fs.ptr, HIP_SYMBOL(scratch));
        if (e == hipSuccess && !fs.done)
            e = hipEventCreateWithFlags(&fs.done, hipEventDisableTiming);
        ctx->fixed_scratch = e == hipSuccess ? fs.ptr : nullptr;
    } else {
        e = hipErrorInvalidDevice;
    }
    if (e != hipSuccess || !ctx->fixed_scratch) {
        (void)hipFree(ctx->d_result);
        (void)hipHostFree(ctx->h_small);
        delete ctx;
        return HF_ERR_HIP;
    }
    *out = ctx;
    return HF_OK;
}

int hf_ctx_destroy(hf_ctx *ctx)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_OK;
    (void)hipDeviceSynchronize(); // work on any stream may still use the blocks freed below
    if (ctx->scratch)
        (void)hipFree(ctx->scratch);
    if (ctx->scratch_done)
        (void)hipEventDestroy(ctx->scratch_done);
    if (ctx->arena)
        (void)hipFree(ctx->arena);
    for (auto &b : ctx->blob_cache) {
        if (b.free_after)
            (void)hipEventDestroy(b.free_after);
        (void)hipFree(b.d);
    }
    if (ctx->up_stream)
        (void)hipStreamDestroy(ctx->up_stream);
    if (ctx->down_stream)
        (void)hipStreamDestroy(ctx->down_stream);
    for (auto &t : ctx->timed)
        for (auto &ev : t.ring) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
    for (auto &ps : ctx->pinned) {
        if (ps.h)
            (void)hipHostFree(ps.h);
        if (ps.done)
            (void)hipEventDestroy(ps.done);
    }
    if (ctx->d_result)
        (void)hipFree(ctx->d_result);
    if (ctx->h_small)
        (void)hipHostFree(ctx->h_small);
    delete ctx;
    return HF_OK;
}

int hf_ctx_set_stream(hf_ctx *ctx, void *stream)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    ctx->stream = static_cast<hipStream_t>(stream);
    return HF_OK;
}

int hf_ctx_synchronize(hf_ctx *ctx)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_ERR_BAD_ARG;
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

const char *hf_ctx_encode_kernel(const hf_ctx *ctx) { return ctx ? ctx->encode_kernel : ""; }
const char *hf_ctx_decode_kernel(const hf_ctx *ctx) { return ctx ? ctx->decode_kernel : ""; }

int hf_ctx_kernel_ms(hf_ctx *ctx, int kind, double *sum_ms, uint64_t *launches)
{
    DeviceGuard on(ctx);
    if (!ctx || kind < 0 || kind >= HF_TIMED_KINDS || !sum_ms || !launches)
        return HF_ERR_BAD_ARG;
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hf_ctx::Timed &t = ctx->timed[kind];
    double sum = 0;
    for (size_t i = 0; i < t.pending; ++i) {
        float ms = 0;
        HF_HIP(ctx, hipEventElapsedTime(&ms, t.ring[i].first, t.ring[i].second));
        sum += ms;
    }
    *sum_ms = sum;
    *launches = t.pending;
    t.pending = 0;
    return HF_OK;
}

const char *hf_last_error(const hf_ctx *ctx) { return ctx ? ctx->err : "no context"; }

int hf_ctx_set_option(hf_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name)
        return HF_ERR_BAD_ARG;
    if (!std::strcmp(name, "enc_mode") && (value == 0 || value == 1))
        ctx->enc_mode = (int)value;
    else if (!std::strcmp(name, "hist_grid") && value >= 0 && value <= 65536)
        ctx->hist_grid = (int)value;
    else if (!std::strcmp(name, "timeout_ms") && value > 0 && value <= 60000)
        ctx->timeout_ms = (int)value;
    else if (!std::strcmp(name, "host_piece_mib") && value >= 1 && value <= 4096)
        ctx->host_piece_mib = (int)value;
    else if (!std::strcmp(name, "kernel_timing"))
        ctx->kernel_timing = value != 0;
    else if (!std::strcmp(name, "dec_trace")) // device address, 128 bytes per segment; 0 = off
        ctx->dec_trace = reinterpret_cast<uint64_t *>((uintptr_t)value);
    else if (!std::strcmp(name, "debug") && value >= 0 && value < 4096)
        ctx->debug = (int)value;
    else
        return fail(ctx, HF_ERR_BAD_ARG, "unknown option or value out of range: %s", name);
    return HF_OK;
}

// ---------------------------------------------------------------- histogram

int hf_histogram_async(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist)
{
    DeviceGuard on(ctx);
    if (!ctx || !d_hist || (!d_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram: null pointer");
    if (n >= (1ull << 40))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram: n must be below 2^40");
    HF_HIP(ctx, hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), ctx->stream));
    if (n == 0)
        return HF_OK;
    const uint64_t per_block = (uint64_t)HIST_THREADS * 16 * HIST_VECS;
    uint64_t want = (n + per_block - 1) / per_block;
    uint64_t cap = ctx->hist_grid > 0 ? (uint64_t)ctx->hist_grid : (uint64_t)ctx->num_cus * 2;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
    auto kern = n >= HIST_STREAM_MIN_BYTES ? hist_kernel<true> : hist_kernel<false>;
    HF_TIMED_LAUNCH(ctx, HF_TIMED_HISTOGRAM,
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(HIST_THREADS), 0, ctx->stream,
                                       d_in, n, reinterpret_cast<unsigned long long *>(d_hist)));
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

uint64_t hf_batch_hist_bytes(uint64_t n)
{
    return (n + PLAN_BATCH - 1) / PLAN_BATCH * 256 * sizeof(uint32_t);
}

int hf_histogram_batches_async(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t *d_hist,
                               uint32_t *d_batch_hist)
{
    DeviceGuard on(ctx);
    if (!ctx || !d_hist || (!d_in && n) || (!d_batch_hist && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_batches: null pointer");
    if (n >= (1ull << 40))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_batches: n must be below 2^40");
    if ((uintptr_t)d_batch_hist & 15)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_batches: d_batch_hist must be 16-byte aligned");
    HF_HIP(ctx, hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), ctx->stream));
    if (n == 0)
        return HF_OK;
    const uint64_t n_batches = (n + PLAN_BATCH - 1) / PLAN_BATCH;
    uint64_t cap = ctx->hist_grid > 0 ? (uint64_t)ctx->hist_grid : (uint64_t)ctx->num_cus * 2;
    const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min(n_batches, cap));
    auto kern = ((uintptr_t)d_in & 15) != 0  ? hist_batches_kernel<false, false>
                : n >= HIST_STREAM_MIN_BYTES ? hist_batches_kernel<true, true>
                                             : hist_batches_kernel<true, false>;
    HF_TIMED_LAUNCH(ctx, HF_TIMED_HISTOGRAM,
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(HIST_THREADS), 0, ctx->stream, d_in,
                                       n, reinterpret_cast<unsigned long long *>(d_hist),
                                       d_batch_hist, n_batches));
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

int hf_histogram(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, uint64_t hist[256])
{
    if (!ctx || !hist)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram: null pointer");
    DeviceGuard on(ctx);
    int rc = scratch_acquire(ctx, 256 * sizeof(uint64_t));
    if (rc != HF_OK)
        return rc;
    uint64_t *d_hist = static_cast<uint64_t *>(ctx->scratch);
    rc = hf_histogram_async(ctx, d_in, n, d_hist);
    if (rc != HF_OK)
        return rc;
    HF_HIP(ctx, hipMemcpyAsync(hist, d_hist, 256 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->scratch_busy = false; // everything that used the block has completed
    return HF_OK;
}

// -------------------------------------------------------------------- model

int hf_model_create(hf_ctx *ctx, const uint64_t codes[256], const uint64_t lens[256],
                    hf_model **out)
{
    DeviceGuard on(ctx);
    if (!ctx || !codes || !lens || !out)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_model_create: null pointer");
    *out = nullptr;
    uint64_t maxlen = 0;
    for (int i = 0; i < 256; ++i) {
        if (lens[i] > 64)
            return fail(ctx, HF_ERR_CODELEN, "symbol %d: code length %llu > 64", i,
                        (unsigned long long)lens[i]);
        if (lens[i] > HF_MAX_CODELEN)
            return fail(ctx, HF_ERR_UNSUPPORTED, "symbol %d: code length %llu > %u", i,
                        (unsigned long long)lens[i], HF_MAX_CODELEN);
        if (lens[i] < 64 && (codes[i] >> lens[i]) != 0)
            return fail(ctx, HF_ERR_BAD_CODEBOOK, "symbol %d: code does not fit %llu bits", i,
                        (unsigned long long)lens[i]);
        maxlen = std::max(maxlen, lens[i]);
    }
    if (maxlen == 0)
        return fail(ctx, HF_ERR_EMPTY, "model has no symbol with a code");
    bool speculative = false;
    const int group = hf::pick_merge_width(lens, &speculative);
    uint64_t enc[256];
    uint32_t enc2[512];
    for (int i = 0; i < 256; ++i) {
        enc[i] = (codes[i] << 6) | lens[i];
        const bool fits = lens[i] <= 32;
        enc2[2 * i + 0] = fits ? (uint32_t)codes[i] : 0u;
        enc2[2 * i + 1] = fits ? (uint32_t)lens[i] : (ENC_LONG_FLAG | (uint32_t)lens[i]);
    }
    uint64_t keys[256], m = 0;
    uint8_t syms[256], klens[256];
    int rc = hf::model_table(codes, lens, keys, syms, klens, &m);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_model_create: cannot build the key table");
    hf::DecodeLut lut;
    rc = hf::build_decode_lut(keys, syms, klens, m, maxlen, &lut);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_model_create: cannot build the decode table");
    hf_model *mod = new (std::nothrow) hf_model();
    if (!mod)
        return HF_ERR_INTERNAL;
    mod->ctx = ctx;
    mod->maxlen = (uint32_t)maxlen;
    mod->has_encode = true;
    mod->bytewise = std::all_of(lens, lens + 256, [](uint64_t l) { return l == 8; });
    mod->enc_group = group;
    mod->enc_spec = speculative;
    rc = upload_model(ctx, mod, enc, enc2, lut, keys, syms, klens, m);
    if (rc != HF_OK) {
        if (mod->d_blob) // taken from the cache or allocated before the failing step
            (void)hipFree(mod->d_blob);
        delete mod;
        return rc;
    }
    *out = mod;
    return HF_OK;
}

int hf_model_from_table(hf_ctx *ctx, const uint64_t *keys, const uint8_t *syms,
                        const uint8_t *klens, uint64_t m, uint64_t sentinel, hf_model **out)
{
    DeviceGuard on(ctx);
    if (!ctx || !keys || !syms || !klens || !out)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_model_from_table: null pointer");
    *out = nullptr;
    if (m == 0 || m > 256)
        return fail(ctx, HF_ERR_EMPTY, "hf_model_from_table: %llu entries", (unsigned long long)m);
    if (sentinel == 0 || sentinel > HF_MAX_SENTINEL)
        return fail(ctx, HF_ERR_UNSUPPORTED, "sentinel %llu outside 1..%u",
                    (unsigned long long)sentinel, HF_MAX_SENTINEL);
    for (uint64_t i = 0; i < m; ++i) {
        if (i && keys[i] <= keys[i - 1])
            return fail(ctx, HF_ERR_BAD_ARG, "keys must be strictly ascending");
        if (sentinel < 64 && (keys[i] >> sentinel) != 0)
            return fail(ctx, HF_ERR_BAD_CODEBOOK, "key %llu wider than the sentinel",
                        (unsigned long long)i);
        if (klens[i] > HF_MAX_CODELEN)
            return fail(ctx, HF_ERR_UNSUPPORTED, "entry %llu: length above %u",
                        (unsigned long long)i, HF_MAX_CODELEN);
    }
    hf::DecodeLut lut;
    int rc = hf::build_decode_lut(keys, syms, klens, m, sentinel, &lut);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_model_from_table: cannot build the decode table");
    hf_model *mod = new (std::nothrow) hf_model();
    if (!mod)
        return HF_ERR_INTERNAL;
    mod->ctx = ctx;
    mod->maxlen = (uint32_t)sentinel;
    mod->has_encode = false;
    rc = upload_model(ctx, mod, nullptr, nullptr, lut, keys, syms, klens, m);
    if (rc != HF_OK) {
        if (mod->d_blob) // taken from the cache or allocated before the failing step
            (void)hipFree(mod->d_blob);
        delete mod;
        return rc;
    }
    *out = mod;
    return HF_OK;
}

int hf_model_destroy(hf_model *model)
{
    if (!model)
        return HF_OK;
    DeviceGuard on(model->ctx);
    if (model->d_blob) {
        // Kernels that read the tables may still be queued.  The block is kept
        // for the next model, whose upload is ordered behind them on the same
        // stream; only tables that other streams have used need a full wait.
        if (model->foreign)
            (void)hipDeviceSynchronize();
        if (model->ctx->blob_cache.size() < 4) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(ev, model->use_stream) != hipSuccess) {
                if (ev)
                    (void)hipEventDestroy(ev);
                ev = nullptr;
                (void)hipDeviceSynchronize();
            }
            model->ctx->blob_cache.push_back({model->d_blob, model->blob_bytes, ev});
        } else {
            (void)hipStreamSynchronize(model->use_stream);
            (void)hipFree(model->d_blob);
        }
    }
    delete model;
    return HF_OK;
}

// ------------------------------------------------------------------- encode

uint64_t hf_encode_bound(const hf_model *model, uint64_t n, uint32_t lead_bits)
{
    const uint64_t maxlen = model ? model->maxlen : 64;
    const uint64_t bytes = (lead_bits + n * maxlen + 7) / 8;
    return (bytes + 3) & ~3ull; // whole 32-bit words keep every store on the fast path
}

int hf_gap_layout(uint64_t n, uint32_t chunk_syms, uint64_t *n_chunks, uint64_t *n_segs)
{
    if (!valid_chunk_syms(chunk_syms) || !n_chunks || !n_segs)
        return HF_ERR_BAD_ARG;
    *n_chunks = (n + chunk_syms - 1) / chunk_syms;
    *n_segs = (*n_chunks + HF_SEG_CHUNKS - 1) / HF_SEG_CHUNKS;
    return HF_OK;
}

} // extern "C"

namespace {

// hf_encode_async and hf_encode_planned_async (d_batch_hist != null)
// `fixed_scratch` / `fixed_bytes`: a caller-supplied scratch block that replaces
// the context's growable one (then nothing is allocated and no kernel is timed).
int encode_launch(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                  uint32_t lead_bits, const uint32_t *d_batch_hist, uint8_t *d_out,
                  uint64_t out_cap, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                  uint64_t *d_seg_off, uint64_t *d_result, void *fixed_scratch = nullptr,
                  size_t fixed_bytes = 0)
{
    DeviceGuard on(ctx);
    if (!ctx || !model || !d_result || (!d_in && n) || (!d_out && out_cap))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: null pointer");
    if (!model->has_encode)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: model was built for decoding only");
    if (int rc0 = model_on_stream(ctx, model))
        return rc0;
    if (lead_bits > 7)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: lead_bits must be 0..7");
    if ((uintptr_t)d_out & 3)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: d_out must be 4-byte aligned");
    if ((d_chunk_bits == nullptr) != (d_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: pass both gap arrays or neither");
    if (d_chunk_bits && !valid_chunk_syms(chunk_syms))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: chunk_syms must be 64, 128 or 256");
    if (n >= (1ull << 45))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode: n too large");
    if (!d_batch_hist || n == 0) // (the planned path's first kernel clears the block itself)
        HF_HIP(ctx, hipMemsetAsync(d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    if (n == 0) {
        const uint64_t bytes = (lead_bits + 7) / 8;
        if (bytes > out_cap)
            return fail(ctx, HF_ERR_CAPACITY, "hf_encode: output buffer too small");
        if (bytes)
            HF_HIP(ctx, hipMemsetAsync(d_out, 0, bytes, ctx->stream));
        return HF_OK;
    }
    if (model->bytewise && lead_bits == 0 && !(ctx->debug & 32)) {
        // 8 bits per symbol throughout: a byte map, no scratch, no scan (enc_bytes_kernel)
        if (d_batch_hist)
            HF_HIP(ctx, hipMemsetAsync(d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
        const uint64_t spans = (n >> 4) / ((uint64_t)ENCB_VECS * ENCB_THREADS);
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spans, (uint64_t)ctx->num_cus * 2));
        const uint32_t cs = d_chunk_bits ? chunk_syms : 64;
        // (non-temporal accesses from 256 MiB on, like the counting pass)
        auto kern = n >= (256ull << 20) ? enc_bytes_kernel<true> : enc_bytes_kernel<false>;
        ctx->encode_kernel = "enc_bytes_kernel";
        if (!fixed_scratch)
            HF_TIMED_LAUNCH(ctx, HF_TIMED_ENCODE,
                            hipLaunchKernelGGL(kern, dim3(grid), dim3(ENCB_THREADS), 0, ctx->stream,
                                               d_in, n, model->d_enc, d_out, out_cap, cs,
                                               d_chunk_bits, d_seg_off, d_result));
        else
            hipLaunchKernelGGL(kern, dim3(grid), dim3(ENCB_THREADS), 0, ctx->stream, d_in, n,
                               model->d_enc, d_out, out_cap, cs, d_chunk_bits, d_seg_off, d_result);
        HF_HIP(ctx, hipGetLastError());
        return HF_OK;
    }
    const uint64_t n_tiles64 = (n + ENC_TILE - 1) / ENC_TILE;
    const uint32_t n_tiles = (uint32_t)n_tiles64;

    EncParams p{};
    p.in = d_in;
    p.n = n;
    p.tab = model->d_enc;
    p.tab2 = model->d_enc2;
    p.len8 = model->d_len8;
    p.out = d_out;
    p.out_cap = out_cap;
    p.lead_bits = lead_bits;
    p.aligned16 = ((uintptr_t)d_in & 15) == 0;
    p.out_aligned16 = ((uintptr_t)d_out & 15) == 0;
    p.chunk_syms = d_chunk_bits ? chunk_syms : 64;
    p.chunk_bits = d_chunk_bits;
    p.seg_off = d_seg_off;
    p.n_chunks = d_chunk_bits ? (n + chunk_syms - 1) / chunk_syms : 0;
    p.result = d_result;
    p.n_tiles = n_tiles;
    p.timeout_ticks = (uint64_t)ctx->timeout_ms * 100000ull;
    p.debug = (uint32_t)ctx->debug & 31u;
#ifdef HF_ENC_TRACE
    if (out_cap < ((uint64_t)n_tiles / ENC_BATCH + 3) * ENC_TRACE_BYTES)
#endif
        p.debug &= ~16u; // the phase trace overwrites the tail of the output buffer

    size_t need = ((size_t)n_tiles + 4) * sizeof(uint64_t);
    // Planned encode of a model with longer codes whose groups of four (or two)
    // rarely exceed 31 bits: the one-lookup kernel, which lists the tiles it
    // cannot finish for enc_fix_tiles_kernel (16 bytes per tile behind the plan
    // arrays; not with the fixed scratch block, which is too small for the list)
    const bool spec_stream = d_batch_hist && !fixed_scratch && model->maxlen > ENC_STREAM_MAXLEN &&
                             model->enc_group >= 2 && !(ctx->debug & 512);
    if (spec_stream)
        need += 16 + (size_t)n_tiles * 16;
    void *work = fixed_scratch;
    if (fixed_scratch) {
        if (need > fixed_bytes)
            return fail(ctx, HF_ERR_UNSUPPORTED,
                        "hf_encode: %llu symbols need %zu bytes of scratch, the fixed block has %zu",
                        (unsigned long long)n, need, fixed_bytes);
    } else {
        int rc = scratch_acquire(ctx, need);
        if (rc != HF_OK)
            return rc;
        work = ctx->scratch;
    }
    // codes merged per append, and whether the table flags codes too long for that
    const int g = model->enc_group;
    const bool flagged = model->enc_spec;
    // every merged group stays below 32 bits: the carry sink applies
    const bool carry = !flagged && g > 1 && (uint32_t)g * model->maxlen <= 31u;
    const uint32_t n_batches = (n_tiles + ENC_BATCH - 1) / ENC_BATCH;
    // batches whose tiles are all complete run the lean instantiation; the
    // ragged last batch (and everything, if the input is unaligned) the general one
    const uint32_t n_lean = p.aligned16 ? (uint32_t)(n / ((uint64_t)ENC_TILE * ENC_BATCH)) : 0u;
    const bool planned = d_batch_hist != nullptr;
    const bool chained = !planned && ctx->enc_mode != 0;
    if (planned) {
        // scratch: group offsets (+ total) | start inside the group | bit total, per batch
        const uint32_t n_groups = (n_batches + ENC_WAVES - 1) / ENC_WAVES;
        uint64_t *off = static_cast<uint64_t *>(work);
        uint32_t *local = reinterpret_cast<uint32_t *>(off + n_groups + 1);
        uint32_t *tot = local + n_batches;
        if (spec_stream) {
            p.fail_count = reinterpret_cast<uint32_t *>(
                (reinterpret_cast<uintptr_t>(tot + n_batches) + 15) & ~(uintptr_t)15);
            p.fail_list = p.fail_count + 4;
        }
        hipLaunchKernelGGL(enc_plan_bits_kernel, dim3(n_groups), dim3(ENC_THREADS), 0, ctx->stream,
                           d_batch_hist, model->d_len8, off, local, tot, n_batches, d_result,
                           p.fail_count);
        hipLaunchKernelGGL(enc_scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, off,
                           n_groups, d_result);
        p.plan_off = off;
        p.plan_local = local;
        p.plan_tot = tot;
    } else if (!chained) {
        p.tile_off = static_cast<uint64_t *>(work);
        hipLaunchKernelGGL(enc_tile_bits_kernel, dim3(n_tiles), dim3(ENC_THREADS), 0, ctx->stream,
                           p);
        hipLaunchKernelGGL(enc_scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, p.tile_off,
                           n_tiles, d_result);
    } else {
        HF_HIP(ctx, hipMemsetAsync(work, 0, need, ctx->stream));
        p.desc = static_cast<unsigned long long *>(work);
        p.ticket = reinterpret_cast<uint32_t *>(p.desc + n_batches);
        p.abort_flag = p.ticket + 2;
    }
    for (int pass = 0; pass < 2; ++pass) {
        const bool lean = pass == 0;
        const uint32_t grid = lean ? n_lean : n_batches - n_lean;
        if (grid == 0)
            continue;
        p.first_batch = lean ? 0u : n_lean;
        void (*kern)(EncParams) = nullptr;
#define HF_PICK(G)                                                                            \
    (flagged ? (lean ? enc_pack_kernel<G, true, true, false>                                  \
                     : enc_pack_kernel<G, true, false, false>)                                \
             : carry ? (lean ? enc_pack_kernel<G, false, true, (G > 1)>                       \
                             : enc_pack_kernel<G, false, false, (G > 1)>)                     \
                     : (lean ? enc_pack_kernel<G, false, true, false>                         \
                             : enc_pack_kernel<G, false, false, false>))
        kern = g == 4 ? HF_PICK(4) : (g == 2 ? HF_PICK(2) : HF_PICK(1));
#undef HF_PICK
        uint32_t grid_wgs = grid;
        // complete planned batches of a model whose codes have at most
        // ENC_STREAM_MAXLEN bits, whatever merge width the model was given for the
        // other kernel, or of one with longer codes that are merged four at a time:
        // one lookup per symbol (debug bit 512, timing experiments: the other
        // kernel instead)
        const bool stream = planned && lean && !(ctx->debug & 512) &&
                            (model->maxlen <= ENC_STREAM_MAXLEN || spec_stream);
        if (stream) {
            if (spec_stream && g == 4)
                kern = enc_pack_stream_kernel<4, false, true>;
            else if (spec_stream)
                kern = enc_pack_stream_kernel<2, false, true>;
            else if (model->maxlen * 4u <= 22u) // merged codes of at most 22 bits: one register per group
                kern = enc_pack_stream_kernel<4, true, false>;
            else if (model->maxlen * 4u <= 31u)
                kern = enc_pack_stream_kernel<4, false, false>;
            else // (two codes of at most ENC_STREAM_MAXLEN bits always are)
                kern = enc_pack_stream_kernel<2, true, false>;
            // Batches per workgroup: one, or two for the largest inputs.  Short runs
            // keep the workgroups that are resident together on neighbouring parts of
            // the input and of the stream and let the hardware balance the load: runs
            // that spread the batches over 3 or 4 workgroups per CU from the start (16
            // batches each at 1 GiB, 256 at 16 GiB) were 7 to 10 % slower.
            p.wg_batches = grid >= 65536u ? 2u : 1u;
            p.n_wg_batches = grid;
            grid_wgs = (grid + p.wg_batches - 1) / p.wg_batches;
        }
        if (lean)
            ctx->encode_kernel = stream ? "enc_pack_stream_kernel" : "enc_pack_kernel";
        // (only the launch over the complete batches is timed: the ragged rest is one
        // workgroup -- together with the kernel that encodes the tiles the speculative
        // one-lookup kernel left out, as a rule none or a handful: its time belongs to
        // the encode of the complete batches)
        const bool fix = lean && spec_stream && stream;
        auto launch = [&] {
            hipLaunchKernelGGL(kern, dim3(grid_wgs), dim3(ENC_THREADS), 0, ctx->stream, p);
            if (fix)
                hipLaunchKernelGGL(enc_fix_tiles_kernel, dim3(ENC_FIX_GRID), dim3(ENC_THREADS), 0,
                                   ctx->stream, p);
        };
        if (lean && !fixed_scratch)
            HF_TIMED_LAUNCH(ctx, HF_TIMED_ENCODE, launch());
        else
            launch();
        if (chained && lean) // the follow-up launch numbers its own workgroups from zero again
            p.ticket += 1;
    }
    HF_HIP(ctx, hipGetLastError());
    if (!fixed_scratch)
        return scratch_release(ctx);
    return HF_OK;
}

} // namespace

extern "C" {

int hf_encode_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                    uint32_t lead_bits, uint8_t *d_out, uint64_t out_cap, uint32_t chunk_syms,
                    uint16_t *d_chunk_bits, uint64_t *d_seg_off, uint64_t *d_result)
{
    return encode_launch(ctx, model, d_in, n, lead_bits, nullptr, d_out, out_cap, chunk_syms,
                         d_chunk_bits, d_seg_off, d_result);
}

int hf_encode_planned_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                            uint32_t lead_bits, const uint32_t *d_batch_hist, uint8_t *d_out,
                            uint64_t out_cap, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                            uint64_t *d_seg_off, uint64_t *d_result)
{
    if (!d_batch_hist && n)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_planned: null batch histograms");
    if ((uintptr_t)d_batch_hist & 15)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_planned: d_batch_hist must be 16-byte aligned");
    return encode_launch(ctx, model, d_in, n, lead_bits, d_batch_hist, d_out, out_cap, chunk_syms,
                         d_chunk_bits, d_seg_off, d_result);
}

int hf_encode_noalloc_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
                            uint32_t lead_bits, const uint32_t *d_batch_hist, uint8_t *d_out,
                            uint64_t out_cap, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                            uint64_t *d_seg_off, uint64_t *d_result)
{
    if (!ctx)
        return HF_ERR_BAD_ARG;
    if (n > HF_NOALLOC_MAX_SYMBOLS)
        return fail(ctx, HF_ERR_UNSUPPORTED,
                    "hf_encode_noalloc: %llu symbols exceed the fixed scratch block's limit of %llu",
                    (unsigned long long)n, (unsigned long long)HF_NOALLOC_MAX_SYMBOLS);
    if ((uintptr_t)d_batch_hist & 15)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_noalloc: d_batch_hist must be 16-byte aligned");
    // Every context and stream on this GPU shares the one block: a call on another
    // stream than the block's last user waits for that user's event first.
    // the kernels get the block's device address, never the array's name
    DeviceGuard on(ctx);
    std::lock_guard<std::mutex> lock(g_fixed_scratch_mutex);
    FixedScratch &fs = g_fixed_scratch[ctx->device];
    if (fs.used && fs.last_stream != ctx->stream)
        HF_HIP(ctx, hipStreamWaitEvent(ctx->stream, fs.done, 0));
    const int rc = encode_launch(ctx, model, d_in, n, lead_bits, d_batch_hist, d_out, out_cap,
                                 chunk_syms, d_chunk_bits, d_seg_off, d_result, ctx->fixed_scratch,
                                 HF_NOALLOC_SCRATCH_BYTES);
    // (also after a failed call: some of its kernels may have been queued)
    if (hipEventRecord(fs.done, ctx->stream) == hipSuccess) {
        fs.used = true;
        fs.last_stream = ctx->stream;
    }
    return rc;
}

int hf_encode(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t n,
              uint32_t lead_bits, uint8_t *d_out, uint64_t out_cap, uint32_t chunk_syms,
              uint16_t *d_chunk_bits, uint64_t *d_seg_off, uint64_t *total_bits)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_ERR_BAD_ARG;
    int rc = hf_encode_async(ctx, model, d_in, n, lead_bits, d_out, out_cap, chunk_syms,
                             d_chunk_bits, d_seg_off, ctx->d_result);
    if (rc != HF_OK)
        return rc;
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (total_bits)
        *total_bits = res[HF_RES_TOTAL_BITS];
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

int hf_encoder_write_value(hf_ctx *ctx, const uint8_t *d_in, uint64_t n, const uint64_t codes[256],
                           const uint64_t lens[256], uint64_t state[3], uint64_t *ret)
{
    if (!ctx || !codes || !lens || !state || !ret || (!d_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encoder_write_value: null pointer");
    if (n > HF_WRITE_VALUE_MAX)
        return fail(ctx, HF_ERR_UNSUPPORTED, "hf_encoder_write_value: more than %llu symbols",
                    (unsigned long long)HF_WRITE_VALUE_MAX);
    DeviceGuard on(ctx);
    const size_t bytes = (512 + 3 + 2) * sizeof(uint64_t);
    int rc = scratch_acquire(ctx, bytes);
    if (rc != HF_OK)
        return rc;
    uint64_t host[512 + 3 + 2];
    std::memcpy(host, codes, 256 * sizeof(uint64_t));
    std::memcpy(host + 256, lens, 256 * sizeof(uint64_t));
    std::memcpy(host + 512, state, 3 * sizeof(uint64_t));
    host[515] = host[516] = 0;
    uint64_t *d = static_cast<uint64_t *>(ctx->scratch);
    HF_HIP(ctx, hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(enc_automaton_kernel, dim3(1), dim3(64), 0, ctx->stream, d_in, n, d, d + 256,
                       d + 512, d + 515);
    HF_HIP(ctx, hipGetLastError());
    HF_HIP(ctx, hipMemcpyAsync(host + 512, d + 512, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->scratch_busy = false;
    if (host[516])
        return fail(ctx, HF_ERR_CODELEN, "Codelen > 64");
    std::memcpy(state, host + 512, 3 * sizeof(uint64_t));
    *ret = host[515];
    return HF_OK;
}

// ------------------------------------------------------------------- decode

static int fill_dec_params(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in,
                           uint64_t in_bytes, uint64_t bit_base, uint64_t n_out, uint8_t *d_out,
                           uint64_t *d_result, DecParams *p)
{
    if (!ctx || !model || !d_result || (!d_out && n_out) || (!d_in && in_bytes))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: null pointer");
    if ((uintptr_t)d_in & 3)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: d_in must be 4-byte aligned");
    if (int rc0 = model_on_stream(ctx, model))
        return rc0;
    p->in = d_in;
    p->in_bytes = in_bytes;
    p->bit_base = bit_base;
    p->n_out = n_out;
    p->out = d_out;
    p->result = d_result;
    p->lut = model->d_lut;
    p->lut_entries = model->lut_entries;
    p->w1 = model->w1;
    p->sentinel = model->sentinel;
    p->keys = model->d_keys;
    p->ksyms = model->d_ksyms;
    p->klens = model->d_klens;
    p->m = model->m;
    p->out_aligned16 = ((uintptr_t)d_out & 15) == 0;
    p->trace = ctx->dec_trace;
    p->in_aligned16 = ((uintptr_t)d_in & 15) == 0;
    return HF_OK;
}

int hf_decode_async(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                    uint64_t bit_base, uint64_t n_out, uint32_t chunk_syms,
                    const uint16_t *d_chunk_bits, const uint64_t *d_seg_off, uint8_t *d_out,
                    uint64_t *d_result)
{
    DeviceGuard on(ctx);
    DecParams p{};
    int rc = fill_dec_params(ctx, model, d_in, in_bytes, bit_base, n_out, d_out, d_result, &p);
    if (rc != HF_OK)
        return rc;
    if (!valid_chunk_syms(chunk_syms))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: chunk_syms must be 64, 128 or 256");
    if (n_out && (!d_chunk_bits || !d_seg_off))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: gap arrays missing (use hf_decode_serial)");
    HF_HIP(ctx, hipMemsetAsync(d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    if (n_out == 0)
        return HF_OK;
    p.chunk_syms = chunk_syms;
    p.chunk_bits = d_chunk_bits;
    p.seg_off = d_seg_off;
    p.n_chunks = (n_out + chunk_syms - 1) / chunk_syms;
    const uint64_t n_segs = (p.n_chunks + HF_SEG_CHUNKS - 1) / HF_SEG_CHUNKS;
    // LDS: table + compressed image (sized for 8 bits per symbol; denser
    // segments read from global memory instead) + padded output image.
    const uint32_t lut_bytes = (model->lut_entries * 2u + 15u) & ~15u;
    // symbols decoded per refill test: the reader guarantees 33 valid bits
    const int spr = model->sentinel <= 8 ? 4 : (model->sentinel <= 16 ? 2 : 1);
    void (*kern)(DecParams) = nullptr;
    size_t lds = 0;
    // symbols per window rebuild of the register-output kernel: K * sentinel <= 64
    // (or the model's speculative K).  Its instantiations for tables without
    // escape entries start at K = 5; a table whose codes are all short enough
    // for the first level but whose sentinel leaves k < 5 (few, long, aligned
    // codes) takes the K = 1..4 instantiations, which exist with the escape
    // test compiled in (the test is then simply never taken), and the
    // BitReader kernel for the larger chunk sizes.
    const uint32_t k = 64u / std::max<uint32_t>(model->sentinel, 1u);
    const bool short_k = k < 5 && model->dec_spec_k <= k;
    const bool reg128 = chunk_syms == 128 && !model->lut_escapes && !short_k;
    const bool reg256 = chunk_syms == 256 && !model->lut_escapes && !short_k;
    // two symbols per lookup: the model has a pair table and a window holds >= 6 codes
    const uint32_t kk = std::max(k, model->dec_spec_k);
    const bool pairs = chunk_syms == 64 && model->pair_entries != 0 && kk >= 6 &&
                       !(ctx->debug & 128);
    bool swz = model->regular_chunks;
    if (ctx->debug & 64) // timing experiments: the other image layout
        swz = !swz;
    if ((chunk_syms == 64 || reg128 || reg256) && !(ctx->debug & 256)) {
        // register-output kernel: LDS holds the tables and the segment's span of
        // the stream only (the output leaves through registers).  The image is
        // sized for the density of this stream -- the larger of what the model's
        // code lengths imply and what the buffer actually holds, plus a quarter
        // -- and never above 8 bits per symbol + slack; a segment that needs
        // more reads from global memory instead (slower, same result).
        const uint32_t worst = HF_SEG_CHUNKS * chunk_syms / 4 + 16;
        const double actual_bits = n_out ? 8.0 * (double)in_bytes / (double)n_out : 8.0;
        const double bits = std::min(8.0, std::max(model->mean_bits, actual_bits) * 1.25 + 0.25);
        uint32_t cap = (uint32_t)(bits * HF_SEG_CHUNKS * chunk_syms / 32.0) + 64;
        cap = std::min(worst, (cap + 3u) & ~3u);
        p.img_cap = cap;
        p.istg_words = cap + cap / 32 + 4; // padded slots, see SwzWords
        lds = lut_bytes + (size_t)(p.istg_words + 4) * 4;
        if (pairs) {
            p.pair = model->d_pair;
            p.pair_entries = model->pair_entries;
            lds += (size_t)model->pair_entries * 4;
            kern = kk >= 16 ? (swz ? dec_kernel_reg<64, 16, true, true, true, true>
                                   : dec_kernel_reg<64, 16, true, false, true, true>)
                            : (kk >= 8 ? (swz ? dec_kernel_reg<64, 8, true, true, true, true>
                                              : dec_kernel_reg<64, 8, true, false, true, true>)
                                       : (swz ? dec_kernel_reg<64, 6, true, true, true, true>
                                              : dec_kernel_reg<64, 6, true, false, true, true>));
        } else if (reg256) // 64 output registers per lane, two work-groups per CU
            kern = swz
                       ? (k >= 6 ? dec_kernel_reg<256, 6, false, true, false>
                                 : dec_kernel_reg<256, 5, false, true, false>)
                       : (k >= 6 ? dec_kernel_reg<256, 6, false, false, false>
                                 : dec_kernel_reg<256, 5, false, false, false>);
        else if (reg128)
            kern = swz
                       ? (k >= 6 ? dec_kernel_reg<128, 6, false, true, false>
                                 : dec_kernel_reg<128, 5, false, true, false>)
                       : (k >= 6 ? dec_kernel_reg<128, 6, false, false, false>
                                 : dec_kernel_reg<128, 5, false, false, false>);
        else if (model->dec_spec_k > k) {
            const uint32_t sk = model->dec_spec_k;
            if (model->lut_escapes)
                kern = sk >= 16 ? dec_kernel_reg<64, 16, true, false, true>
                                : (sk >= 8 ? dec_kernel_reg<64, 8, true, false, true>
                                           : (sk == 7 ? dec_kernel_reg<64, 7, true, false, true>
                                                      : dec_kernel_reg<64, 6, true, false, true>));
            else
                kern = sk >= 16 ? dec_kernel_reg<64, 16, false, false, true>
                                : (sk >= 8 ? dec_kernel_reg<64, 8, false, false, true>
                                           : (sk == 7 ? dec_kernel_reg<64, 7, false, false, true>
                                                      : dec_kernel_reg<64, 6, false, false, true>));
        } else if (model->lut_escapes || short_k)
            kern = k >= 4 ? dec_kernel_reg<64, 4, true, false, false>
                          : (k == 3 ? dec_kernel_reg<64, 3, true, false, false>
                                    : (k == 2 ? dec_kernel_reg<64, 2, true, false, false>
                                              : dec_kernel_reg<64, 1, true, false, false>));
        else if (swz)
            kern = k >= 8 ? dec_kernel_reg<64, 8, false, true, false>
                          : (k >= 6 ? dec_kernel_reg<64, 6, false, true, false>
                                    : dec_kernel_reg<64, 5, false, true, false>);
        else
            kern = k >= 8 ? dec_kernel_reg<64, 8, false, false, false>
                          : (k >= 6 ? dec_kernel_reg<64, 6, false, false, false>
                                    : dec_kernel_reg<64, 5, false, false, false>);
    } else {
        p.istg_words = HF_SEG_CHUNKS * chunk_syms / 4 + 16;
        const uint32_t ostg_words = HF_SEG_CHUNKS * (chunk_syms / 4 + 1);
        lds = lut_bytes + (size_t)(p.istg_words + ostg_words + 4) * 4;
        if (model->lut_escapes)
            kern = spr == 4 ? dec_kernel<4, true>
                            : (spr == 2 ? dec_kernel<2, true> : dec_kernel<1, true>);
        else
            kern = spr == 4 ? dec_kernel<4, false>
                            : (spr == 2 ? dec_kernel<2, false> : dec_kernel<1, false>);
    }
    uint32_t grid = (uint32_t)n_segs;
    const uint32_t threads = DEC_THREADS;
    ctx->decode_kernel = p.img_cap ? (pairs ? "dec_kernel_reg (pairs)" : "dec_kernel_reg") : "dec_kernel";
    // pair tables, plain image, aligned buffers: two segments per work-group, two
    // look-up chains per lane (the tables are staged once for both)
    if (p.img_cap && pairs && !swz && p.in_aligned16 && p.out_aligned16 && n_segs >= 2 &&
        !(ctx->debug & 1024)) {
        kern = kk >= 16 ? dec_kernel_pair2<16> : (kk >= 8 ? dec_kernel_pair2<8> : dec_kernel_pair2<6>);
        p.istg_words = p.img_cap + 4; // (plain image: no padding slots)
        lds = lut_bytes + (size_t)model->pair_entries * 4 + (size_t)(2 * p.istg_words + 8) * 4;
        grid = (uint32_t)((n_segs + 1) / 2);
        ctx->decode_kernel = "dec_kernel_pair2";
    }
    if (lds > 160 * 1024 - 64)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode: chunk_syms %u needs %zu bytes of LDS",
                    chunk_syms, lds);
    HF_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HF_TIMED_LAUNCH(ctx, HF_TIMED_DECODE,
                    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, ctx->stream, p));
    HF_HIP(ctx, hipGetLastError());
    return HF_OK;
}

int hf_decode(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
              uint64_t bit_base, uint64_t n_out, uint32_t chunk_syms,
              const uint16_t *d_chunk_bits, const uint64_t *d_seg_off, uint8_t *d_out)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_ERR_BAD_ARG;
    int rc = hf_decode_async(ctx, model, d_in, in_bytes, bit_base, n_out, chunk_syms, d_chunk_bits,
                             d_seg_off, d_out, ctx->d_result);
    if (rc != HF_OK)
        return rc;
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

int hf_decode_serial(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                     uint64_t bit_base, uint64_t n_out, uint8_t *d_out)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_ERR_BAD_ARG;
    DecParams p{};
    int rc = fill_dec_params(ctx, model, d_in, in_bytes, bit_base, n_out, d_out, ctx->d_result, &p);
    if (rc != HF_OK)
        return rc;
    if (bit_base != 0)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_serial: bit_base must be 0");
    HF_HIP(ctx, hipMemsetAsync(ctx->d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
    if (n_out) {
        hipLaunchKernelGGL(dec_serial_kernel, dim3(1), dim3(64), 0, ctx->stream, p);
        HF_HIP(ctx, hipGetLastError());
    }
    uint64_t res[HF_RES_WORDS];
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

} // extern "C"

namespace {

// hf_build_gap_array, the same for a window of a longer stream, and either
// with the symbols decoded in the last walk.
//   first_start  bit of d_in[0] at which the first code starts (0..255)
//   bit_base     stream bit of bit 0 of d_in[0]: the segment offsets are absolute
//   whole        the window reaches the end of the stream: n_want symbols are
//                expected (HF_ERR_DECODE if it holds fewer).  Otherwise the
//                codes that start in the window's last 2048 bits are left for
//                the next window (they may run past its end), and as many
//                whole segments as the rest holds, at most n_want symbols,
//                are laid out; *n_done receives that count (possibly 0).
//   d_chunk_bits, d_seg_off   receive the gap array, or both null: none wanted
//   d_out        receives the symbols, or null: none wanted
//   *end_bit     bit (from bit 0 of d_in[0]) behind the last of those symbols
int gap_build(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
              uint32_t first_start, uint64_t bit_base, bool whole, uint64_t n_want,
              uint32_t chunk_syms, uint16_t *d_chunk_bits, uint64_t *d_seg_off, uint8_t *d_out,
              uint64_t *n_done, uint64_t *end_bit)
{
    GapParams g{};
    const bool gaps = d_chunk_bits != nullptr;
    // (d.out is only written when d_out is given; any non-null value passes the shared
    // argument checks)
    int rc = fill_dec_params(ctx, model, d_in, in_bytes, 0, n_want,
                             d_out ? d_out : reinterpret_cast<uint8_t *>(ctx->d_result),
                             ctx->d_result, &g.d);
    if (rc != HF_OK)
        return rc;
    if (in_bytes == 0 || in_bytes >= (1ull << 58))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: empty or oversized stream");
    g.d.chunk_syms = chunk_syms;
    // Codes taken from one 64-bit window without a look at the bounds: what a window
    // always holds (64 / sentinel), or -- complete prefix codes only -- what it nearly
    // always holds under the model's own code lengths (the decoder's speculative window
    // width): steps that took more are caught and done again.  The sum of a window's
    // lengths must fit a byte.  And whether the table has entries that need more than
    // the first level.
    g.safe_w = std::max(1u, 64u / std::max(1u, g.d.sentinel));
    uint32_t per_window = g.safe_w;
    if (model->dec_spec_k > per_window)
        per_window = std::min<uint32_t>(model->dec_spec_k, 8u);
    while (per_window > g.safe_w && per_window * g.d.sentinel > 255u)
        --per_window;
    const bool esc = model->lut_escapes;
    struct GapKernels {
        void (*sync)(GapParams);
        void (*emit)(GapParams);
        void (*decode)(GapParams);
    };
#define HF_GAP_KERNELS(W, E) GapKernels{gap_sync_kernel<W, E>, gap_emit_kernel<W, E>, gap_decode_kernel<W, E>}
    GapKernels kern =
        per_window >= 8 ? (esc ? HF_GAP_KERNELS(8, true) : HF_GAP_KERNELS(8, false))
        : per_window >= 6 ? (esc ? HF_GAP_KERNELS(6, true) : HF_GAP_KERNELS(6, false))
        : per_window >= 5 ? (esc ? HF_GAP_KERNELS(5, true) : HF_GAP_KERNELS(5, false))
        : per_window >= 4 ? HF_GAP_KERNELS(4, true)
        : per_window >= 3 ? HF_GAP_KERNELS(3, true)
        : per_window >= 2 ? HF_GAP_KERNELS(2, true)
                          : HF_GAP_KERNELS(1, true);
#undef HF_GAP_KERNELS
    // steep codes (a lane's 512 bits hold a dozen steps of sixteen codes and more): the
    // last walk stores eight steps at a time
    if (per_window >= 8 && model->mean_bits <= 3.0)
        kern.decode = esc ? gap_decode_kernel<8, true, 8> : gap_decode_kernel<8, false, 8>;
    g.total_bits = in_bytes * 8;
    g.n_sub = (g.total_bits + GAP_SUB_BITS - 1) / GAP_SUB_BITS;
    g.n_wg = (g.n_sub + GAP_OWN - 1) / GAP_OWN;
    g.n_chunks = (n_want + chunk_syms - 1) / chunk_syms; // (upper bound until n_out is known)
    g.chunk_bits = d_chunk_bits;
    g.seg_off = d_seg_off;
    g.first_start = first_start;
    g.bit_base = bit_base;
    if (g.n_wg >= (1ull << 31))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: stream too long");

    // scratch: start_off | exit_off | cnt | wg_in | wg_used | changed | wg_base | start32 | blk_total
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t o_start = take(g.n_sub), o_exit = take(g.n_sub), o_cnt = take(g.n_sub * 4);
    const size_t o_in = take(g.n_wg * 4), o_used = take(g.n_wg * 4), o_chg = take(4);
    const size_t o_base = take((g.n_wg + 1) * 8), o_s32 = take(gaps ? (g.n_chunks + 1) * 4 : 4);
    const uint32_t scan_blocks = (uint32_t)((g.n_wg + GAP_SCAN_BLOCK - 1) / GAP_SCAN_BLOCK);
    const size_t o_blk = take((size_t)scan_blocks * 4);
    rc = scratch_acquire(ctx, off);
    if (rc != HF_OK)
        return rc;
    uint8_t *b = static_cast<uint8_t *>(ctx->scratch);
    g.start_off = b + o_start;
    g.exit_off = b + o_exit;
    g.cnt = reinterpret_cast<uint32_t *>(b + o_cnt);
    g.wg_in = reinterpret_cast<uint32_t *>(b + o_in);
    g.wg_used = reinterpret_cast<uint32_t *>(b + o_used);
    g.changed = reinterpret_cast<uint32_t *>(b + o_chg);
    g.wg_base = reinterpret_cast<uint64_t *>(b + o_base);
    g.blk_total = reinterpret_cast<uint32_t *>(b + o_blk);
    g.start32 = reinterpret_cast<uint32_t *>(b + o_s32);
    HF_HIP(ctx, hipMemsetAsync(ctx->d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));

    const uint32_t lut_bytes = (g.d.lut_entries * 2u + 15u) & ~15u;
    const size_t lds = lut_bytes + (size_t)GAP_IMG_SLOTS * 4 + GAP_THREADS * 4;
    const dim3 wgs((uint32_t)g.n_wg), one_per_wg((uint32_t)((g.n_wg + GAP_THREADS - 1) / GAP_THREADS));
    auto last_walk = [&]() { // gap array and / or symbols from the final starts
        if (gaps) {
            HF_TIMED_LAUNCH(ctx, HF_TIMED_GAP_WALK,
                            hipLaunchKernelGGL(kern.emit, wgs, dim3(GAP_THREADS), lds, ctx->stream, g));
            hipLaunchKernelGGL(gap_diff_kernel,
                               dim3((uint32_t)((g.n_chunks + GAP_THREADS - 1) / GAP_THREADS)),
                               dim3(GAP_THREADS), 0, ctx->stream, g);
        }
        if (d_out)
            HF_TIMED_LAUNCH(ctx, HF_TIMED_GAP_WALK,
                            hipLaunchKernelGGL(kern.decode, wgs, dim3(GAP_THREADS), lds, ctx->stream, g));
    };
    // The first launch lets every work-group assume its first start (from a walk
    // of its predecessor's last sub-sequence); the check behind it finds those that
    // assumed wrong, and they run again with the value their predecessor computed.
    // Every round fixes at least the first work-group whose first start was wrong,
    // so n_wg + 1 rounds always suffice; real streams need none or one.  For a whole
    // stream the sums and the last walk are queued behind the check at once and the
    // host looks at everything together: when the check fails (rarely) they run again.
    g.first = 1;
    uint64_t total_syms = 0;
    uint64_t res[HF_RES_WORDS] = {0};
    for (uint64_t round = 0;; ++round) {
        if (round > g.n_wg + 1)
            return fail(ctx, HF_ERR_INTERNAL, "hf_build_gap_array: no fixed point");
        HF_TIMED_LAUNCH(ctx, HF_TIMED_GAP_SYNC,
                        hipLaunchKernelGGL(kern.sync, wgs, dim3(GAP_THREADS), lds, ctx->stream, g));
        HF_HIP(ctx, hipGetLastError());
        g.first = 0;
        HF_HIP(ctx, hipMemsetAsync(g.changed, 0, 4, ctx->stream));
        hipLaunchKernelGGL(gap_verify_kernel, one_per_wg, dim3(GAP_THREADS), 0, ctx->stream, g);
        hipLaunchKernelGGL(gap_wgsum_kernel, wgs, dim3(GAP_THREADS), 0, ctx->stream, g);
        hipLaunchKernelGGL(gap_scan_local_kernel, dim3(scan_blocks), dim3(GAP_THREADS), 0, ctx->stream, g);
        hipLaunchKernelGGL(gap_scan_add_kernel, dim3(scan_blocks), dim3(GAP_THREADS), 0, ctx->stream, g);
        uint64_t *h = ctx->h_small; // [0..3] result block, [4] changed, [5] symbols in the stream
        if (whole) {
            HF_HIP(ctx, hipMemsetAsync(ctx->d_result, 0, HF_RES_WORDS * sizeof(uint64_t), ctx->stream));
            last_walk();
            HF_HIP(ctx, hipMemcpyAsync(h, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
        }
        h[4] = 0;
        HF_HIP(ctx, hipMemcpyAsync(h + 4, g.changed, 4, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(ctx, hipMemcpyAsync(h + 5, g.wg_base + g.n_wg, 8, hipMemcpyDeviceToHost, ctx->stream));
        HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(res, h, sizeof(res));
        total_syms = h[5];
        if (!(uint32_t)h[4])
            break;
    }
    HF_HIP(ctx, hipGetLastError());
    uint64_t n_out = n_want;
    if (whole) {
        ctx->scratch_busy = false;
        if (total_syms < n_want)
            return fail(ctx, HF_ERR_DECODE,
                        "hf_build_gap_array: the stream holds %llu symbols, fewer than n_out = %llu",
                        (unsigned long long)total_syms, (unsigned long long)n_want);
        *n_done = n_out;
        *end_bit = res[HF_RES_END_BIT];
        return flags_to_status(ctx, res[HF_RES_FLAGS]);
    }
    {
        // symbols whose code starts before the window's last two sub-sequences
        uint64_t safe = 0;
        if (g.total_bits >= 4 * (uint64_t)GAP_SUB_BITS) {
            const uint64_t j_lim = g.total_bits / GAP_SUB_BITS - 2; // first sub-sequence left out
            const uint64_t wg = j_lim / GAP_OWN;
            uint32_t cnt[GAP_THREADS];
            const uint64_t first_sub = wg * GAP_OWN, n_cnt = j_lim - first_sub;
            HF_HIP(ctx, hipMemcpyAsync(&safe, g.wg_base + wg, 8, hipMemcpyDeviceToHost, ctx->stream));
            if (n_cnt)
                HF_HIP(ctx, hipMemcpyAsync(cnt, g.cnt + first_sub, n_cnt * 4, hipMemcpyDeviceToHost,
                                           ctx->stream));
            HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (uint64_t i = 0; i < n_cnt; ++i)
                safe += cnt[i];
        }
        const uint64_t seg_syms = (uint64_t)HF_SEG_CHUNKS * chunk_syms;
        n_out = std::min(n_want, safe) / seg_syms * seg_syms;
    }
    *n_done = n_out;
    *end_bit = first_start;
    if (n_out == 0) {
        ctx->scratch_busy = false;
        return HF_OK;
    }
    g.d.n_out = n_out;
    g.n_chunks = (n_out + chunk_syms - 1) / chunk_syms;
    last_walk();
    HF_HIP(ctx, hipGetLastError());
    HF_HIP(ctx, hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->scratch_busy = false;
    *end_bit = res[HF_RES_END_BIT];
    return flags_to_status(ctx, res[HF_RES_FLAGS]);
}

} // namespace

extern "C" {

int hf_build_gap_array(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                       uint64_t n_out, uint32_t chunk_syms, uint16_t *d_chunk_bits,
                       uint64_t *d_seg_off)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_ERR_BAD_ARG;
    if (!valid_chunk_syms(chunk_syms))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: chunk_syms must be 64, 128 or 256");
    if (n_out && (!d_chunk_bits || !d_seg_off))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_build_gap_array: null output");
    if (n_out == 0) {
        DecParams unused{};
        return fill_dec_params(ctx, model, d_in, in_bytes, 0, 0, nullptr, ctx->d_result, &unused);
    }
    uint64_t n_done = 0, end_bit = 0;
    return gap_build(ctx, model, d_in, in_bytes, 0, 0, true, n_out, chunk_syms, d_chunk_bits,
                     d_seg_off, nullptr, &n_done, &end_bit);
}

int hf_decode_bare(hf_ctx *ctx, const hf_model *model, const uint8_t *d_in, uint64_t in_bytes,
                   uint64_t n_out, uint8_t *d_out)
{
    DeviceGuard on(ctx);
    if (!ctx)
        return HF_ERR_BAD_ARG;
    if (n_out && !d_out)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_bare: null output");
    if (n_out == 0) {
        DecParams unused{};
        return fill_dec_params(ctx, model, d_in, in_bytes, 0, 0, nullptr, ctx->d_result, &unused);
    }
    uint64_t n_done = 0, end_bit = 0;
    return gap_build(ctx, model, d_in, in_bytes, 0, 0, true, n_out, 64, nullptr, nullptr, d_out,
                     &n_done, &end_bit);
}

// ------------------------------------------------------- host-buffer forms

} // extern "C"

namespace {

// hf_encode_host for an input of several pieces: all uploads are queued on the
// upload stream at once; every piece is encoded as a stream of its own that
// starts with the bits the previous pieces left in their last byte
// (lead_bits), so the pieces concatenate bytewise and only the byte two pieces
// share is OR-merged on the host.  A piece's length is needed before the next
// can start; that wait only idles the (barely loaded) compute stream, uploads
// and downloads keep running.  Two output buffers take turns.
int encode_host_pieces(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t n,
                       uint32_t lead_bits, uint8_t *h_out, uint64_t out_cap, uint32_t chunk_syms,
                       uint16_t *h_chunk_bits, uint64_t *h_seg_off, uint64_t *total_bits,
                       uint64_t piece)
{
    if (!ctx->up_stream)
        HF_HIP(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
    if (!ctx->down_stream)
        HF_HIP(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
    const uint64_t pieces = (n + piece - 1) / piece;
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits)
        hf_gap_layout(n, chunk_syms, &n_chunks, &n_segs);
    const uint64_t seg_syms = (uint64_t)HF_SEG_CHUNKS * chunk_syms;
    const uint64_t cap_piece = align_up(hf_encode_bound(model, piece, 7) + 16, 256);
    // arena: input | two stream buffers | chunk_bits | seg_off | result blocks
    const size_t off_out = align_up(n + 16, 256);
    const size_t off_cb = off_out + 2 * cap_piece;
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    const size_t off_res = align_up(off_so + n_segs * 8 + 16, 256);
    int rc = ensure_arena(ctx, off_res + pieces * HF_RES_WORDS * sizeof(uint64_t) + 16);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    uint16_t *d_cb = reinterpret_cast<uint16_t *>(a + off_cb);
    uint64_t *d_so = reinterpret_cast<uint64_t *>(a + off_so);
    uint64_t *d_res = reinterpret_cast<uint64_t *>(a + off_res);
    std::vector<hipEvent_t> ev(2 * pieces, nullptr); // [2i] uploaded, [2i + 1] downloaded
    std::vector<uint8_t> first(pieces, 0);           // a piece's first byte when it is shared ...
    std::vector<uint8_t> shared(pieces, 0);          // ... with the last byte of its predecessor
    std::vector<uint64_t> base(pieces, 0);           // stream bit at which a piece's own bits start
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == HF_OK)
            rc = fail(ctx, HF_ERR_HIP, "hf_encode_host: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (auto &e : ev)
        if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event"))
            break;
    for (uint64_t i = 0; i < pieces && rc == HF_OK; ++i) {
        const uint64_t at = i * piece, len = std::min(piece, n - at);
        hip_ok(hipMemcpyAsync(a + at, h_in + at, len, hipMemcpyHostToDevice, ctx->up_stream),
               "upload");
        hip_ok(hipEventRecord(ev[2 * i], ctx->up_stream), "record");
    }
    uint64_t running = lead_bits; // bits of the stream so far, the caller's lead bits included
    for (uint64_t i = 0; i < pieces && rc == HF_OK; ++i) {
        const uint64_t at = i * piece, len = std::min(piece, n - at);
        const uint32_t lead = (uint32_t)(running & 7u);
        uint8_t *d_out = a + off_out + (i & 1) * cap_piece;
        hip_ok(hipStreamWaitEvent(ctx->stream, ev[2 * i], 0), "wait");
        if (i >= 2) // the buffer is free once the piece before the last has been downloaded
            hip_ok(hipStreamWaitEvent(ctx->stream, ev[2 * (i - 2) + 1], 0), "wait");
        if (rc != HF_OK)
            break;
        uint64_t *res_i = d_res + i * HF_RES_WORDS;
        rc = hf_encode_async(ctx, model, a + at, len, lead, d_out, cap_piece, chunk_syms,
                             h_chunk_bits ? d_cb + at / chunk_syms : nullptr,
                             h_seg_off ? d_so + at / seg_syms : nullptr, res_i);
        if (rc != HF_OK)
            break;
        uint64_t res[HF_RES_WORDS] = {0};
        if (!hip_ok(hipMemcpyAsync(res, res_i, sizeof(res), hipMemcpyDeviceToHost, ctx->stream),
                    "result") ||
            !hip_ok(hipStreamSynchronize(ctx->stream), "synchronize"))
            break;
        if ((rc = flags_to_status(ctx, res[HF_RES_FLAGS])) != HF_OK)
            break;
        const uint64_t bits = res[HF_RES_TOTAL_BITS];
        const uint64_t nbytes = (lead + bits + 7) / 8, dst = running / 8;
        if (dst + nbytes > out_cap) {
            rc = fail(ctx, HF_ERR_CAPACITY, "hf_encode_host: output buffer too small");
            break;
        }
        // the byte shared with the previous piece is merged after all downloads
        const uint64_t skip = (lead != 0 && i > 0 && nbytes > 0) ? 1 : 0;
        shared[i] = (uint8_t)skip;
        if (skip)
            hip_ok(hipMemcpyAsync(&first[i], d_out, 1, hipMemcpyDeviceToHost, ctx->down_stream),
                   "download");
        if (nbytes > skip)
            hip_ok(hipMemcpyAsync(h_out + dst + skip, d_out + skip, nbytes - skip,
                                  hipMemcpyDeviceToHost, ctx->down_stream), "download");
        hip_ok(hipEventRecord(ev[2 * i + 1], ctx->down_stream), "record");
        base[i] = running - lead;
        running += bits;
    }
    if (rc == HF_OK && h_chunk_bits && n_chunks) {
        hip_ok(hipMemcpyAsync(h_chunk_bits, d_cb, n_chunks * 2, hipMemcpyDeviceToHost,
                              ctx->down_stream), "download");
        hip_ok(hipMemcpyAsync(h_seg_off, d_so, n_segs * 8, hipMemcpyDeviceToHost, ctx->down_stream),
               "download");
    }
    (void)hipStreamSynchronize(ctx->up_stream);
    (void)hipStreamSynchronize(ctx->stream);
    hip_ok(hipStreamSynchronize(ctx->down_stream), "synchronize");
    for (auto &e : ev)
        if (e)
            (void)hipEventDestroy(e);
    if (rc != HF_OK)
        return rc;
    for (uint64_t i = 0; i < pieces; ++i)
        if (shared[i]) // base[i] / 8 is the byte the piece shares with its predecessor
            h_out[base[i] / 8] |= first[i];
    if (h_seg_off) { // a piece counted its segment offsets from its own first byte
        const uint64_t segs_per_piece = piece / seg_syms;
        for (uint64_t sgm = 0; sgm < n_segs; ++sgm)
            h_seg_off[sgm] += base[sgm / segs_per_piece];
    }
    if (total_bits)
        *total_bits = running - lead_bits;
    return HF_OK;
}

// hf_decode_host for a stream with gap arrays that spans several pieces of
// `piece_segs` segments: the gap arrays are on the device already (queued on the
// context's stream); piece i needs the stream bytes up to the start of piece
// i + 1 (plus what the readers run ahead), which the upload stream delivers in
// order; the decode of piece i waits for its upload, the download stream for
// its decode.  Every piece has its own result block; their flags are merged.
int decode_host_pieces(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t in_bytes,
                       uint64_t n_out, uint32_t chunk_syms, const uint64_t *h_seg_off,
                       const uint16_t *d_cb, const uint64_t *d_so, uint8_t *d_in, uint8_t *d_out,
                       uint8_t *h_out, uint64_t piece_segs, uint64_t pieces, uint64_t *d_res)
{
    if (!ctx->up_stream)
        HF_HIP(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
    if (!ctx->down_stream)
        HF_HIP(ctx, hipStreamCreateWithFlags(&ctx->down_stream, hipStreamNonBlocking));
    const uint64_t seg_syms = (uint64_t)HF_SEG_CHUNKS * chunk_syms;
    const uint64_t n_segs = (n_out + seg_syms - 1) / seg_syms;
    std::vector<hipEvent_t> ev(2 * pieces, nullptr);
    int rc = HF_OK;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == HF_OK)
            rc = fail(ctx, HF_ERR_HIP, "hf_decode_host: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    for (auto &e : ev)
        if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "event"))
            break;
    uint64_t uploaded = 0;
    for (uint64_t i = 0; i < pieces && rc == HF_OK; ++i) {
        const uint64_t s0 = i * piece_segs, s1 = std::min(n_segs, s0 + piece_segs);
        // bytes this piece may touch: up to the next piece's first bit, plus run-ahead
        uint64_t need = s1 < n_segs ? std::min<uint64_t>(in_bytes, (h_seg_off[s1] / 8 + 64 + 15) & ~15ull)
                                    : in_bytes;
        need = std::max(need, uploaded);
        if (need > uploaded)
            hip_ok(hipMemcpyAsync(d_in + uploaded, h_in + uploaded, need - uploaded,
                                  hipMemcpyHostToDevice, ctx->up_stream), "upload");
        uploaded = need;
        hip_ok(hipEventRecord(ev[2 * i], ctx->up_stream), "record");
        hip_ok(hipStreamWaitEvent(ctx->stream, ev[2 * i], 0), "wait");
        if (rc != HF_OK)
            break;
        const uint64_t sym0 = s0 * seg_syms;
        const uint64_t n_piece = std::min(n_out, s1 * seg_syms) - sym0;
        rc = hf_decode_async(ctx, model, d_in, in_bytes, 0, n_piece, chunk_syms,
                             d_cb + s0 * HF_SEG_CHUNKS, d_so + s0, d_out + sym0,
                             d_res + i * HF_RES_WORDS);
        if (rc != HF_OK)
            break;
        hip_ok(hipEventRecord(ev[2 * i + 1], ctx->stream), "record");
        hip_ok(hipStreamWaitEvent(ctx->down_stream, ev[2 * i + 1], 0), "wait");
        hip_ok(hipMemcpyAsync(h_out + sym0, d_out + sym0, n_piece, hipMemcpyDeviceToHost,
                              ctx->down_stream), "download");
    }
    std::vector<uint64_t> res(pieces * HF_RES_WORDS, 0);
    (void)hipStreamSynchronize(ctx->up_stream);
    hip_ok(hipStreamSynchronize(ctx->stream), "synchronize");
    hip_ok(hipStreamSynchronize(ctx->down_stream), "synchronize");
    if (rc == HF_OK)
        hip_ok(hipMemcpy(res.data(), d_res, res.size() * sizeof(uint64_t), hipMemcpyDeviceToHost),
               "results");
    for (auto &e : ev)
        if (e)
            (void)hipEventDestroy(e);
    if (rc != HF_OK)
        return rc;
    uint64_t flags = 0;
    for (uint64_t i = 0; i < pieces; ++i)
        flags |= res[i * HF_RES_WORDS + HF_RES_FLAGS];
    return flags_to_status(ctx, flags);
}

} // namespace

extern "C" {

int hf_histogram_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n, uint64_t hist[256])
{
    DeviceGuard on(ctx);
    if (!ctx || !hist || (!h_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_histogram_host: null pointer");
    int rc = ensure_arena(ctx, n + 16);
    if (rc != HF_OK)
        return rc;
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(ctx->arena, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    return hf_histogram(ctx, static_cast<const uint8_t *>(ctx->arena), n, hist);
}

int hf_encode_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t n,
                   uint32_t lead_bits, uint8_t *h_out, uint64_t out_cap, uint32_t chunk_syms,
                   uint16_t *h_chunk_bits, uint64_t *h_seg_off, uint64_t *total_bits)
{
    DeviceGuard on(ctx);
    if (!ctx || !model || (!h_in && n) || (!h_out && out_cap))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: null pointer");
    if ((h_chunk_bits == nullptr) != (h_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: pass both gap arrays or neither");
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits && hf_gap_layout(n, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: chunk_syms must be 64, 128 or 256");
    if (lead_bits > 7)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encode_host: lead_bits must be 0..7");
    const uint64_t piece = (uint64_t)ctx->host_piece_mib << 20; // a multiple of every segment size
    if (n > piece)
        return encode_host_pieces(ctx, model, h_in, n, lead_bits, h_out, out_cap, chunk_syms,
                                  h_chunk_bits, h_seg_off, total_bits, piece);
    // arena layout: input | stream | chunk_bits | seg_off, each 256-byte aligned
    const uint64_t bound = std::min<uint64_t>(hf_encode_bound(model, n, lead_bits),
                                              (out_cap + 3) & ~3ull);
    const size_t off_out = align_up(n + 16, 256);
    const size_t off_cb = align_up(off_out + bound + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    const size_t total = off_so + n_segs * 8 + 16;
    int rc = ensure_arena(ctx, total);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(a, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    uint64_t bits = 0;
    rc = hf_encode(ctx, model, a, n, lead_bits, a + off_out, std::min<uint64_t>(bound, out_cap),
                   chunk_syms, h_chunk_bits ? reinterpret_cast<uint16_t *>(a + off_cb) : nullptr,
                   h_seg_off ? reinterpret_cast<uint64_t *>(a + off_so) : nullptr, &bits);
    if (total_bits)
        *total_bits = bits;
    if (rc != HF_OK)
        return rc;
    const uint64_t nbytes = (lead_bits + bits + 7) / 8;
    if (nbytes > out_cap)
        return fail(ctx, HF_ERR_CAPACITY, "hf_encode_host: output buffer too small");
    if (nbytes)
        HF_HIP(ctx, hipMemcpyAsync(h_out, a + off_out, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    if (h_chunk_bits && n_chunks) {
        HF_HIP(ctx, hipMemcpyAsync(h_chunk_bits, a + off_cb, n_chunks * 2, hipMemcpyDeviceToHost,
                                   ctx->stream));
        HF_HIP(ctx, hipMemcpyAsync(h_seg_off, a + off_so, n_segs * 8, hipMemcpyDeviceToHost,
                                   ctx->stream));
    }
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_compress_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n, uint64_t codes[256],
                     uint64_t lens[256], uint8_t *h_out, uint64_t out_cap, uint32_t chunk_syms,
                     uint16_t *h_chunk_bits, uint64_t *h_seg_off, uint64_t *total_bits)
{
    DeviceGuard on(ctx);
    if (!ctx || !codes || !lens || !total_bits || (!h_in && n) || (!h_out && out_cap))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_compress_host: null pointer");
    if ((h_chunk_bits == nullptr) != (h_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_compress_host: pass both gap arrays or neither");
    if (n == 0)
        return fail(ctx, HF_ERR_EMPTY, "hf_compress_host: empty input");
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits && hf_gap_layout(n, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_compress_host: chunk_syms must be 64, 128 or 256");
    // A Huffman code never needs more than the 8 bits per symbol of the plain
    // bytes, so n bytes (rounded up to the store width) always hold the stream.
    // arena: input | histogram | batch histograms | stream | chunk_bits | seg_off
    const uint64_t bound = std::min<uint64_t>((n + 15) & ~15ull, (out_cap + 3) & ~3ull);
    const size_t off_hist = align_up(n + 16, 256);
    const size_t off_bh = align_up(off_hist + 256 * 8, 256);
    const size_t off_out = align_up(off_bh + hf_batch_hist_bytes(n), 256);
    const size_t off_cb = align_up(off_out + bound + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    const size_t total = off_so + n_segs * 8 + 16;
    int rc = ensure_arena(ctx, total);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    uint64_t *d_hist = reinterpret_cast<uint64_t *>(a + off_hist);
    uint32_t *d_bh = reinterpret_cast<uint32_t *>(a + off_bh);
    HF_HIP(ctx, hipMemcpyAsync(a, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    rc = hf_histogram_batches_async(ctx, a, n, d_hist, d_bh);
    if (rc != HF_OK)
        return rc;
    uint64_t hist[256];
    HF_HIP(ctx, hipMemcpyAsync(hist, d_hist, sizeof(hist), hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = hf_build_codebook(hist, 256, codes, lens);
    if (rc != HF_OK)
        return fail(ctx, rc, "hf_compress_host: cannot build the codebook");
    hf_model *model = nullptr;
    rc = hf_model_create(ctx, codes, lens, &model);
    if (rc != HF_OK)
        return rc;
    rc = hf_encode_planned_async(ctx, model, a, n, 0, d_bh, a + off_out,
                                 std::min<uint64_t>(bound, out_cap), chunk_syms,
                                 h_chunk_bits ? reinterpret_cast<uint16_t *>(a + off_cb) : nullptr,
                                 h_seg_off ? reinterpret_cast<uint64_t *>(a + off_so) : nullptr,
                                 ctx->d_result);
    uint64_t res[HF_RES_WORDS] = {0};
    if (rc == HF_OK) {
        // the stream's length is known from the histogram: its download is
        // queued behind the encoder without waiting for the result block
        uint64_t bits = 0;
        for (int i = 0; i < 256; ++i)
            bits += hist[i] * lens[i];
        const uint64_t nbytes = (bits + 7) / 8;
        if (nbytes > out_cap) {
            rc = fail(ctx, HF_ERR_CAPACITY, "hf_compress_host: output buffer too small");
        } else {
            hipError_t e = hipMemcpyAsync(h_out, a + off_out, nbytes, hipMemcpyDeviceToHost,
                                          ctx->stream);
            if (e == hipSuccess && h_chunk_bits && n_chunks) {
                e = hipMemcpyAsync(h_chunk_bits, a + off_cb, n_chunks * 2, hipMemcpyDeviceToHost,
                                   ctx->stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(h_seg_off, a + off_so, n_segs * 8, hipMemcpyDeviceToHost,
                                       ctx->stream);
            }
            if (e == hipSuccess)
                e = hipMemcpyAsync(res, ctx->d_result, sizeof(res), hipMemcpyDeviceToHost,
                                   ctx->stream);
            if (e == hipSuccess)
                e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess)
                rc = fail(ctx, HF_ERR_HIP, "hf_compress_host: %s", hipGetErrorString(e));
            else if ((rc = flags_to_status(ctx, res[HF_RES_FLAGS])) == HF_OK &&
                     res[HF_RES_TOTAL_BITS] != bits)
                rc = fail(ctx, HF_ERR_INTERNAL, "hf_compress_host: stream length mismatch");
            *total_bits = bits;
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    hf_model_destroy(model);
    return rc;
}

int hf_decode_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_in, uint64_t in_bytes,
                   uint64_t n_out, uint32_t chunk_syms, const uint16_t *h_chunk_bits,
                   const uint64_t *h_seg_off, uint8_t *h_out)
{
    DeviceGuard on(ctx);
    if (!ctx || !model || (!h_in && in_bytes) || (!h_out && n_out))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_host: null pointer");
    if ((h_chunk_bits == nullptr) != (h_seg_off == nullptr))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_host: pass both gap arrays or neither");
    uint64_t n_chunks = 0, n_segs = 0;
    if (h_chunk_bits && hf_gap_layout(n_out, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_host: chunk_syms must be 64, 128 or 256");
    const bool rebuild = !h_chunk_bits && n_out > 0;
    if (rebuild) { // no side metadata: the device reconstructs it from the stream
        chunk_syms = 64;
        hf_gap_layout(n_out, chunk_syms, &n_chunks, &n_segs);
    }
    const size_t off_out = align_up(in_bytes + 16, 256);
    const size_t off_cb = align_up(off_out + n_out + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    // with side metadata at hand, a large stream goes through the device in pieces
    // of whole segments whose upload, decode and download overlap
    const uint64_t seg_syms = (uint64_t)HF_SEG_CHUNKS * chunk_syms;
    const uint64_t piece_segs =
        std::max<uint64_t>(1, ((uint64_t)ctx->host_piece_mib << 20) / std::max<uint64_t>(seg_syms, 1));
    const uint64_t pieces = (!rebuild && n_segs > piece_segs) ? (n_segs + piece_segs - 1) / piece_segs : 1;
    const size_t off_res = align_up(off_so + n_segs * 8 + 16, 256);
    const size_t total = off_res + pieces * HF_RES_WORDS * sizeof(uint64_t) + 16;
    int rc = ensure_arena(ctx, total);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    if (in_bytes && pieces == 1)
        HF_HIP(ctx, hipMemcpyAsync(a, h_in, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint16_t *d_cb = reinterpret_cast<uint16_t *>(a + off_cb);
    uint64_t *d_so = reinterpret_cast<uint64_t *>(a + off_so);
    bool serial = false, decoded = false;
    if (rebuild) {
        // the walks that would rebuild the side metadata write the symbols themselves
        uint64_t n_done = 0, end_bit = 0;
        rc = in_bytes ? gap_build(ctx, model, a, in_bytes, 0, 0, true, n_out, chunk_syms, nullptr,
                                  nullptr, a + off_out, &n_done, &end_bit)
                      : HF_ERR_DECODE;
        // a stream that ends early is continued with zero bits, as the reference's
        // drain loop does (src/huffman/decode/mod.rs:219-229): the sequential walk
        if (rc == HF_ERR_DECODE && n_out <= HF_WRITE_VALUE_MAX)
            serial = true;
        else if (rc != HF_OK)
            return rc;
        else
            decoded = true;
    } else if (n_chunks) {
        HF_HIP(ctx, hipMemcpyAsync(d_cb, h_chunk_bits, n_chunks * 2, hipMemcpyHostToDevice,
                                   ctx->stream));
        HF_HIP(ctx, hipMemcpyAsync(d_so, h_seg_off, n_segs * 8, hipMemcpyHostToDevice,
                                   ctx->stream));
    }
    if (serial)
        rc = hf_decode_serial(ctx, model, a, in_bytes, 0, n_out, a + off_out);
    else if (pieces > 1)
        return decode_host_pieces(ctx, model, h_in, in_bytes, n_out, chunk_syms, h_seg_off, d_cb,
                                  d_so, a, a + off_out, h_out, piece_segs, pieces,
                                  reinterpret_cast<uint64_t *>(a + off_res));
    else if (n_out && !decoded)
        rc = hf_decode(ctx, model, a, in_bytes, 0, n_out, chunk_syms, d_cb, d_so, a + off_out);
    if (rc != HF_OK)
        return rc;
    if (n_out)
        HF_HIP(ctx, hipMemcpyAsync(h_out, a + off_out, n_out, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_decode_range_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_win,
                         uint64_t win_bytes, uint64_t bit_base, uint64_t n_out, uint32_t chunk_syms,
                         const uint16_t *h_chunk_bits, const uint64_t *h_seg_off, uint8_t *h_out)
{
    DeviceGuard on(ctx);
    if (!ctx || !model || (!h_win && win_bytes) || (!h_out && n_out))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_range_host: null pointer");
    if (n_out == 0)
        return HF_OK;
    if (!h_chunk_bits || !h_seg_off)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_range_host: gap arrays missing");
    uint64_t n_chunks = 0, n_segs = 0;
    if (hf_gap_layout(n_out, chunk_syms, &n_chunks, &n_segs) != HF_OK)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_range_host: chunk_syms must be 64, 128 or 256");
    if (bit_base & 31u)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_range_host: bit_base must be a multiple of 32");
    const size_t off_out = align_up(win_bytes + 16, 256);
    const size_t off_cb = align_up(off_out + n_out + 16, 256);
    const size_t off_so = align_up(off_cb + n_chunks * 2 + 16, 256);
    int rc = ensure_arena(ctx, off_so + n_segs * 8 + 16);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    if (win_bytes)
        HF_HIP(ctx, hipMemcpyAsync(a, h_win, win_bytes, hipMemcpyHostToDevice, ctx->stream));
    HF_HIP(ctx, hipMemcpyAsync(a + off_cb, h_chunk_bits, n_chunks * 2, hipMemcpyHostToDevice,
                               ctx->stream));
    HF_HIP(ctx, hipMemcpyAsync(a + off_so, h_seg_off, n_segs * 8, hipMemcpyHostToDevice,
                               ctx->stream));
    rc = hf_decode(ctx, model, a, win_bytes, bit_base, n_out, chunk_syms,
                   reinterpret_cast<const uint16_t *>(a + off_cb),
                   reinterpret_cast<const uint64_t *>(a + off_so), a + off_out);
    if (rc != HF_OK)
        return rc;
    HF_HIP(ctx, hipMemcpyAsync(h_out, a + off_out, n_out, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HF_OK;
}

int hf_decode_window_host(hf_ctx *ctx, const hf_model *model, const uint8_t *h_win,
                          uint64_t win_bytes, uint32_t start_bit, uint64_t n_max, int last,
                          uint8_t *h_out, uint64_t *n_decoded, uint64_t *end_bit)
{
    DeviceGuard on(ctx);
    if (!ctx || !model || !n_decoded || !end_bit || (!h_win && win_bytes) || (!h_out && n_max))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_window_host: null pointer");
    if (start_bit > 255)
        return fail(ctx, HF_ERR_BAD_ARG, "hf_decode_window_host: start_bit must be below 256");
    *n_decoded = 0;
    *end_bit = start_bit;
    if (n_max == 0)
        return HF_OK;
    const uint32_t cs = 64; // (whole segments of 64-symbol chunks are what a window hands over)
    const size_t off_out = align_up(win_bytes + 16, 256);
    int rc = ensure_arena(ctx, off_out + n_max + 16);
    if (rc != HF_OK)
        return rc;
    uint8_t *a = static_cast<uint8_t *>(ctx->arena);
    if (win_bytes)
        HF_HIP(ctx, hipMemcpyAsync(a, h_win, win_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint64_t n = 0, end = start_bit;
    rc = win_bytes ? gap_build(ctx, model, a, win_bytes, start_bit, 0, last != 0, n_max, cs, nullptr,
                               nullptr, a + off_out, &n, &end)
                   : HF_ERR_DECODE;
    if (rc == HF_ERR_DECODE && last && n_max <= HF_WRITE_VALUE_MAX && start_bit == 0) {
        // the stream ends early: it is continued with zero bits, as the reference's
        // drain loop does (src/huffman/decode/mod.rs:219-229) -- the sequential walk
        rc = hf_decode_serial(ctx, model, a, win_bytes, 0, n_max, a + off_out);
        if (rc != HF_OK)
            return rc;
        n = n_max;
        end = win_bytes * 8;
    } else if (rc != HF_OK) {
        return rc;
    }
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(h_out, a + off_out, n, hipMemcpyDeviceToHost, ctx->stream));
    HF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_decoded = n;
    *end_bit = end;
    return HF_OK;
}

int hf_encoder_write_value_host(hf_ctx *ctx, const uint8_t *h_in, uint64_t n,
                                const uint64_t codes[256], const uint64_t lens[256],
                                uint64_t state[3], uint64_t *ret)
{
    DeviceGuard on(ctx);
    if (!ctx || (!h_in && n))
        return fail(ctx, HF_ERR_BAD_ARG, "hf_encoder_write_value_host: null pointer");
    if (n > HF_WRITE_VALUE_MAX)
        return fail(ctx, HF_ERR_UNSUPPORTED, "hf_encoder_write_value_host: more than %llu symbols",
                    (unsigned long long)HF_WRITE_VALUE_MAX);
    int rc = ensure_arena(ctx, n + 16);
    if (rc != HF_OK)
        return rc;
    if (n)
        HF_HIP(ctx, hipMemcpyAsync(ctx->arena, h_in, n, hipMemcpyHostToDevice, ctx->stream));
    return hf_encoder_write_value(ctx, static_cast<const uint8_t *>(ctx->arena), n, codes, lens,
                                  state, ret);
}

} // extern "C"
