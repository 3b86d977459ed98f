// im_capi.hip -- the C ABI of include/indelminer_amd.h over the HIP kernels.
// No CPU fallback anywhere: without a gfx950 device every entry point fails.

#include "im_device.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>
#include <unordered_map>
#include <mutex>
#include <atomic>

struct im_ctx {
    int device = -1;
    int n_cu = 0;
    std::atomic<int> expect_len{0};     // im_expect_read_length: beyond kShortRead the long-read pass follows every realign launch
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // reference
    int32_t n_contigs = 0;
    uint8_t* ref_ascii = nullptr;
    uint64_t* ref_pk = nullptr;
    int64_t* d_asc_off = nullptr;
    int64_t* d_pk_off = nullptr;
    int32_t* d_len = nullptr;
    // reusable device workspace for the host-buffer entry points
    void* ws = nullptr;
    size_t ws_bytes = 0;
    // pinned host staging of the host-buffer entry points, and a second stream for their copies
    void* pin = nullptr;
    size_t pin_bytes = 0;
    hipStream_t copy_stream = nullptr, back_stream = nullptr;     // host -> device, device -> host
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    // resident depth array of the current contig (im_depth_build)
    int32_t* depth = nullptr;
    int32_t* depth_sums = nullptr;
    int64_t depth_cap = 0, depth_len = -1;
    // host copies of the reference layout
    std::vector<int64_t> h_asc_off;
    std::vector<int32_t> h_len;
    int64_t ref_total = 0;
    // genome-wide depth / difference array (im_depth_enable): one int32 per byte of ref_ascii
    int32_t* gdepth = nullptr;
    int32_t* gdepth_sums = nullptr;
    std::mutex gb_mu;
    std::unordered_map<void*, int32_t> gb_layout;   // group-by scratch -> the slot count it was initialised (and is carved) for
    std::unordered_map<void*, std::pair<int32_t, int32_t>> fg_layout;   // flush + group-by scratch -> (slots, flushes) it is carved for
    std::vector<int64_t> h_sums_off;    // each contig's own run of tile sums: scans of different contigs may be in flight on different streams
    // read-group -> range[1] table (im_set_insert_ranges), flattened hashtable chains
    void* rg_blob = nullptr;
    im::RgTable rg = {nullptr, 0, 0};
    // the general realign pass (im_realign_any.hip): list of the reads it takes, counters, arena; one call at a time
    std::mutex any_mu;
    int32_t* any_list = nullptr; int32_t any_list_cap = 0;
    int32_t* any_counters = nullptr;
    int32_t* any_arena = nullptr; size_t any_arena_bytes = 0;
};

namespace {

char g_err[512] = "";

void set_err(im_ctx* ctx, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    char* dst = ctx ? ctx->err : g_err;
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_err(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return IM_E_HIP;                                                            \
        }                                                                               \
    } while (0)

int check_params(im_ctx* ctx, const im_params* p)
{
    if (!p) { set_err(ctx, "params is NULL"); return IM_E_ARG; }
    // forceassert((klength > 1) && (klength < 16)), src/indelminer.c:1028
    if (p->klength < 2 || p->klength > 15) { set_err(ctx, "klength %u outside 2..15", p->klength); return IM_E_ARG; }
    if (p->maxdelsize == 0) { set_err(ctx, "maxdelsize must be > 0"); return IM_E_ARG; }
    if (p->numgaps > (1u << 20)) { set_err(ctx, "numgaps=%u", p->numgaps); return IM_E_ARG; }
    return IM_OK;
}

int ensure_ws(im_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return IM_OK;
    if (ctx->ws) { HIP_TRY(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
    bytes = (bytes + (1u << 20) - 1) / (1u << 20) * (1u << 20);
    HIP_TRY(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return IM_OK;
}

int ensure_pin(im_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->pin_bytes) return IM_OK;
    if (ctx->pin) { HIP_TRY(ctx, hipHostFree(ctx->pin)); ctx->pin = nullptr; ctx->pin_bytes = 0; }
    bytes = (bytes + (1u << 20) - 1) / (1u << 20) * (1u << 20);
    HIP_TRY(ctx, hipHostMalloc(&ctx->pin, bytes, hipHostMallocDefault));
    ctx->pin_bytes = bytes;
    return IM_OK;
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

void free_reference(im_ctx* ctx)
{
    if (ctx->ref_ascii) (void)hipFree(ctx->ref_ascii);
    if (ctx->ref_pk) (void)hipFree(ctx->ref_pk);
    if (ctx->d_asc_off) (void)hipFree(ctx->d_asc_off);
    if (ctx->d_pk_off) (void)hipFree(ctx->d_pk_off);
    if (ctx->d_len) (void)hipFree(ctx->d_len);
    ctx->ref_ascii = nullptr; ctx->ref_pk = nullptr; ctx->d_asc_off = nullptr; ctx->d_pk_off = nullptr; ctx->d_len = nullptr;
    ctx->n_contigs = 0;
    if (ctx->gdepth) (void)hipFree(ctx->gdepth);
    if (ctx->gdepth_sums) (void)hipFree(ctx->gdepth_sums);
    ctx->gdepth = nullptr; ctx->gdepth_sums = nullptr;
    ctx->h_asc_off.clear(); ctx->h_len.clear(); ctx->ref_total = 0;
}

}  // namespace

extern "C" {

int im_abi_version(void) { return IM_ABI_VERSION; }

const char* im_last_error(const im_ctx* ctx) { return ctx ? ctx->err : g_err; }

int im_ctx_create(int device, im_ctx** out)
{
    if (!out) { set_err(nullptr, "out is NULL"); return IM_E_ARG; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_err(nullptr, "no HIP device available (%s); this library has no CPU path", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return IM_E_NOGPU;
    }
    if (device < 0 || device >= count) { set_err(nullptr, "device %d out of range (0..%d)", device, count - 1); return IM_E_ARG; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { set_err(nullptr, "hipGetDeviceProperties: %s", hipGetErrorString(e)); return IM_E_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "device %d is %s; the kernels are built for gfx950 only", device, prop.gcnArchName);
        return IM_E_NOGPU;
    }
    im_ctx* ctx = new im_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { set_err(nullptr, "stream create: %s", hipGetErrorString(e)); delete ctx; return IM_E_HIP; }
    *out = ctx;
    return IM_OK;
}

void im_ctx_destroy(im_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_reference(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->depth) (void)hipFree(ctx->depth);
    if (ctx->depth_sums) (void)hipFree(ctx->depth_sums);
    if (ctx->rg_blob) (void)hipFree(ctx->rg_blob);
    if (ctx->any_list) (void)hipFree(ctx->any_list);
    if (ctx->any_counters) (void)hipFree(ctx->any_counters);
    if (ctx->any_arena) (void)hipFree(ctx->any_arena);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (int i = 0; i < 2; i++) {
        if (ctx->ev_in[i]) (void)hipEventDestroy(ctx->ev_in[i]);
        if (ctx->ev_k[i]) (void)hipEventDestroy(ctx->ev_k[i]);
        if (ctx->ev_out[i]) (void)hipEventDestroy(ctx->ev_out[i]);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->back_stream) (void)hipStreamDestroy(ctx->back_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int im_set_reference(im_ctx* ctx, int32_t n_contigs, const char* const* seqs, const int64_t* lens)
{
    if (!ctx) return IM_E_ARG;
    if (n_contigs <= 0 || !seqs || !lens) { set_err(ctx, "bad reference arguments"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    free_reference(ctx);
    // layout: [256 B zero pad][contig 0][>= 64 B zero pad, next start 256-aligned][contig 1]...
    std::vector<int64_t> asc_off(n_contigs), pk_off(n_contigs);
    std::vector<int32_t> len32(n_contigs);
    int64_t pos = 256;
    for (int32_t i = 0; i < n_contigs; i++) {
        if (lens[i] < 0 || lens[i] > 0x7fffff00LL) { set_err(ctx, "contig %d length %lld unsupported", i, (long long)lens[i]); return IM_E_ARG; }
        asc_off[i] = pos;
        pk_off[i] = pos / 4;
        len32[i] = (int32_t)lens[i];
        pos = (int64_t)up256((size_t)(pos + lens[i] + 64));
    }
    const int64_t total = pos + 256;                 // multiple of 256, hence of 32
    HIP_TRY(ctx, hipMalloc((void**)&ctx->ref_ascii, (size_t)total));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->ref_pk, (size_t)(total / 4 + 1024)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_asc_off, sizeof(int64_t) * n_contigs));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_pk_off, sizeof(int64_t) * n_contigs));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_len, sizeof(int32_t) * n_contigs));
    HIP_TRY(ctx, hipMemsetAsync(ctx->ref_ascii, 0, (size_t)total, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->ref_pk, 0, (size_t)(total / 4 + 1024), ctx->stream));
    for (int32_t i = 0; i < n_contigs; i++)
        if (lens[i] > 0)
            HIP_TRY(ctx, hipMemcpyAsync(ctx->ref_ascii + asc_off[i], seqs[i], (size_t)lens[i], hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_asc_off, asc_off.data(), sizeof(int64_t) * n_contigs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pk_off, pk_off.data(), sizeof(int64_t) * n_contigs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_len, len32.data(), sizeof(int32_t) * n_contigs, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, im::launch_pack_reference(ctx->ref_ascii, ctx->ref_pk, total, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_contigs = n_contigs;
    ctx->h_asc_off = asc_off; ctx->h_len = len32; ctx->ref_total = total;
    return IM_OK;
}

static int dev_realign(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, int keep, const int32_t* n_dev, void* stream);

// The general pass: list what is left, size the arena from the longest read and the widest window on the list, work the list off.
// The list's size comes back to the host in between (a stream synchronisation: this pass is for inputs outside what sequencers
// deliver, and cannot be captured into a launch graph); one call at a time per context.
static int realign_any(im_ctx* ctx, const im::RealignArgs& a, int all, hipStream_t stream)
{
    if (a.batch.n <= 0) return IM_OK;
    std::lock_guard<std::mutex> lk(ctx->any_mu);
    if (!ctx->any_counters) HIP_TRY(ctx, hipMalloc((void**)&ctx->any_counters, 4 * sizeof(int32_t)));
    if (a.batch.n > ctx->any_list_cap) {
        if (ctx->any_list) { HIP_TRY(ctx, hipFree(ctx->any_list)); ctx->any_list = nullptr; ctx->any_list_cap = 0; }
        const int32_t cap = a.batch.n < (1 << 16) ? (1 << 16) : a.batch.n;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->any_list, sizeof(int32_t) * (size_t)cap));
        ctx->any_list_cap = cap;
    }
    HIP_TRY(ctx, im::launch_realign_any_pick(a, all, ctx->any_list, ctx->any_counters, stream));
    int32_t h[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->any_counters, sizeof h, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    if (h[0] <= 0) return IM_OK;
    const int32_t max_read = h[1], max_window = h[2];
    // A batch of long reads is small against the chip (13 918 candidates of a 2 x 300 library are 218 waves of 64: less than one per CU
    // with three of its SIMDs idle): a wave then holds FEWER reads -- R = 8 .. 64 lanes with a read each, the others only help in the band
    // searches -- until every SIMD has a wave (a wave's instructions cost the same whatever the number of lanes at work: more waves than
    // SIMDs only add work -- 8 reads per wave took a 2 x 300 batch at -g 61 from 64 to 105 ms).
    const int64_t most = (int64_t)ctx->n_cu * 8, simds = (int64_t)ctx->n_cu * 4;
    int32_t lane_shift = 6;
    // (bands beyond the cooperative form's 32 diagonals: every lane on its own with a read each, as measured best)
    while (a.P.numgaps < 32u && lane_shift > 3 && ((int64_t)h[0] + (1 << (lane_shift - 1)) - 1) >> (lane_shift - 1) <= simds) lane_shift--;
    const size_t budget = (size_t)6 << 30;
    size_t per_wave = im::realign_any_arena_bytes(max_read, max_window, a.P.numgaps, lane_shift, 1);
    int64_t waves = ((int64_t)h[0] + (1 << lane_shift) - 1) >> lane_shift;
    if (waves > most) waves = most;
    if (per_wave * (size_t)waves > budget) waves = (int64_t)(budget / per_wave);
    if (waves < 1) waves = 1;
    const size_t need = per_wave * (size_t)waves;
    if (need > ctx->any_arena_bytes) {
        if (ctx->any_arena) { HIP_TRY(ctx, hipFree(ctx->any_arena)); ctx->any_arena = nullptr; ctx->any_arena_bytes = 0; }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->any_arena, need));
        ctx->any_arena_bytes = need;
    }
    HIP_TRY(ctx, im::launch_realign_any(a, ctx->any_list, ctx->any_counters, ctx->any_arena, max_read, max_window, lane_shift, (int32_t)waves, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));                                   // the list and the arena are free for the next call
    return IM_OK;
}

int im_dev_realign(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, void* stream)
{
    return dev_realign(ctx, params, batch, 0, nullptr, stream);
}
int im_dev_realign_n(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, const int32_t* n_dev, int32_t keep_slots, void* stream)
{
    return dev_realign(ctx, params, batch, keep_slots ? 1 : 0, n_dev, stream);
}
int im_dev_realign_keep(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, void* stream)
{
    return dev_realign(ctx, params, batch, 1, nullptr, stream);
}

static int dev_realign(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, int keep, const int32_t* n_dev, void* stream)
{
    if (!ctx || !batch) return IM_E_ARG;
    int rc = check_params(ctx, params);
    if (rc) return rc;
    if (!ctx->ref_ascii) { set_err(ctx, "im_set_reference has not been called"); return IM_E_ARG; }
    if (batch->n < 0) { set_err(ctx, "negative batch size"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    im::RealignArgs a;
    a.ref.ascii = ctx->ref_ascii; a.ref.pk = reinterpret_cast<const uint8_t*>(ctx->ref_pk);
    a.ref.asc_off = ctx->d_asc_off; a.ref.pk_off = ctx->d_pk_off; a.ref.len = ctx->d_len;
    a.ref.n_contigs = ctx->n_contigs;
    a.batch = *batch;
    a.P = *params;
    a.keep_slots = keep;
    a.n_dev = n_dev;
    a.first = 0;
    // Reads of up to 255 bases (1020 at numgaps == 0) and bands of up to 61 diagonals run in the kernels laid out for them; what
    // those leave IM_ST_UNSUPPORTED -- and every read when the band is wider than a wave -- takes the general pass behind them.
    const int expect = ctx->expect_len.load(std::memory_order_relaxed);
    const bool wide = params->numgaps > (uint32_t)im::kMaxWaveGaps;
    if (!wide) HIP_TRY(ctx, im::launch_realign(a, ctx->n_cu, (hipStream_t)stream));
    if (expect > im::kShortRead && params->numgaps == 0)
        HIP_TRY(ctx, im::launch_realign_long(a, ctx->n_cu, (hipStream_t)stream));
    if (wide || expect > IM_MAX_READ || (params->numgaps > 0 && expect > im::kShortRead))
        return realign_any(ctx, a, wide ? 1 : 0, (hipStream_t)stream);
    return IM_OK;
}

int im_dev_compact_results(im_ctx* ctx, const im_read_result* res, int32_t n, const int32_t* n_dev,
                           int32_t* status, int32_t* slot, im_read_result* compact, int32_t* count, void* stream)
{
    if (!ctx || n < 0 || !count || (n > 0 && (!res || !status || !slot || !compact))) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_compact_results(res, n, n_dev, status, slot, compact, count, ctx->n_cu, (hipStream_t)stream));
    return IM_OK;
}

int im_expect_read_length(im_ctx* ctx, int32_t max_len)
{
    if (!ctx) return IM_E_ARG;
    int cur = ctx->expect_len.load(std::memory_order_relaxed);
    while (max_len > cur && !ctx->expect_len.compare_exchange_weak(cur, max_len)) {}
    return IM_OK;
}

// ---- seam 0: record triage ---------------------------------------------------------------

int im_set_insert_ranges(im_ctx* ctx, int32_t n, const char* const* names_in, const int32_t* range_max)
{
    if (!ctx || n < 0 || (n > 0 && (!names_in || !range_max))) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // add_hashtable prepends to the chain of bin hash & 15 (src/hashtable.c:44-45): the chain order is the
    // REVERSE of the order of insertion
    std::vector<int32_t> bin(n);
    for (int32_t i = 0; i < n; i++) {
        const char* nm = names_in[i];
        const int len = (int)strlen(nm);
        uint32_t h = 5381u;
        for (int k = len - 1; k >= 0; k--) h += (h << 5) + (uint32_t)(int)nm[k];
        bin[i] = (int32_t)(h & 15u);
    }
    const int32_t m = n > 0 ? n : 1;
    std::vector<int32_t> bin_start(20, 0), name_off(m, 0), name_len(m, 0), rmax(m, 0);
    std::vector<uint8_t> names;
    int32_t e = 0;
    for (int b = 0; b < 16; b++) {
        bin_start[b] = e;
        for (int32_t i = n - 1; i >= 0; i--) {
            if (bin[i] != b) continue;
            name_off[e] = (int32_t)names.size();
            name_len[e] = (int32_t)strlen(names_in[i]);
            rmax[e] = range_max[i];
            names.insert(names.end(), names_in[i], names_in[i] + name_len[e] + 1);
            e++;
        }
    }
    bin_start[16] = e;
    std::vector<uint8_t> host((size_t)4 * (20 + 3 * (size_t)m) + names.size() + 8, 0);
    memcpy(host.data(), bin_start.data(), 80);
    memcpy(host.data() + 80, name_off.data(), 4 * (size_t)m);
    memcpy(host.data() + 80 + 4 * (size_t)m, name_len.data(), 4 * (size_t)m);
    memcpy(host.data() + 80 + 8 * (size_t)m, rmax.data(), 4 * (size_t)m);
    if (!names.empty()) memcpy(host.data() + 80 + 12 * (size_t)m, names.data(), names.size());
    const size_t total = (host.size() + 3) / 4 * 4;
    host.resize(total, 0);
    if (ctx->rg_blob) { HIP_TRY(ctx, hipFree(ctx->rg_blob)); ctx->rg_blob = nullptr; }
    HIP_TRY(ctx, hipMalloc(&ctx->rg_blob, total + 256));
    HIP_TRY(ctx, hipMemcpy(ctx->rg_blob, host.data(), total, hipMemcpyHostToDevice));
    ctx->rg.blob = (const uint8_t*)ctx->rg_blob; ctx->rg.n = n; ctx->rg.bytes = (int32_t)total;
    return IM_OK;
}

size_t im_dev_triage_scratch_bytes(int32_t n_records) { return im::triage_scratch_bytes(n_records); }

int im_dev_triage_scratch_init(im_ctx* ctx, int32_t n_records, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || !scratch || scratch_bytes < im::triage_scratch_bytes(n_records)) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    size_t zb = 0;
    const size_t off = im::triage_scratch_zero_offset(n_records, &zb);
    HIP_TRY(ctx, hipMemsetAsync((char*)scratch + off, 0, zb, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_triage(im_ctx* ctx, const im_triage_params* tp, const im_dev_records* recs, const im_dev_cands* out,
                  void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || !tp || !recs || !out) return IM_E_ARG;
    if (!ctx->ref_ascii) { set_err(ctx, "im_set_reference has not been called"); return IM_E_ARG; }
    if (!ctx->rg_blob) { set_err(ctx, "im_set_insert_ranges has not been called"); return IM_E_ARG; }
    if (recs->n < 0 || !out->counters || !out->cand_rec) { set_err(ctx, "bad triage arguments"); return IM_E_ARG; }
    if (scratch_bytes < im::triage_scratch_bytes(recs->n)) { set_err(ctx, "triage scratch too small"); return IM_E_ARG; }
    if (tp->want_depth && !ctx->gdepth) { set_err(ctx, "want_depth without im_depth_enable"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    im::RefDev ref;
    ref.ascii = ctx->ref_ascii; ref.pk = reinterpret_cast<const uint8_t*>(ctx->ref_pk);
    ref.asc_off = ctx->d_asc_off; ref.pk_off = ctx->d_pk_off; ref.len = ctx->d_len; ref.n_contigs = ctx->n_contigs;
    HIP_TRY(ctx, im::launch_triage(ref, ctx->rg, ctx->gdepth, *tp, *recs, *out, scratch, (hipStream_t)stream));
    return IM_OK;
}

// ---- seam 2, streaming form ------------------------------------------------------------------

int im_dev_flush_cut(im_ctx* ctx, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                     int32_t a0, int32_t a1, int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id,
                     uint64_t* cut_word, void* stream)
{
    if (!ctx || !cut_word || flush_id <= 0) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_flush_cut(cls, b1, b2, consumed, a0, a1, b0, b1_end, marker, flush_id, cut_word, nullptr, nullptr, 0, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_flush_cut_rec(im_ctx* ctx, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                         int32_t rec0, int32_t rec1, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap,
                         int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id, uint64_t* cut_word, void* stream)
{
    if (!ctx || !cut_word || flush_id <= 0 || !cand_rec || !n_cand_dev || cand_cap < 0) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_flush_cut(cls, b1, b2, consumed, rec0, rec1, b0, b1_end, marker, flush_id, cut_word, cand_rec, n_cand_dev, cand_cap, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_flush_cuts(im_ctx* ctx, const im_flush_desc* desc_dev, int32_t n_flushes,
                      const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                      const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base, int32_t pe_count, void* stream)
{
    if (!ctx || n_flushes < 0 || !desc_dev || !cand_rec || !n_cand_dev || pe_count < 0) return IM_E_ARG;
    if ((((uintptr_t)cls) | ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)consumed)) & 15u) { set_err(ctx, "im_dev_flush_cuts: the slot arrays must be 16-byte aligned"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_flush_seq(desc_dev, n_flushes, cls, b1, b2, consumed, cand_rec, n_cand_dev, cand_cap, pe_base, pe_count, (hipStream_t)stream));
    return IM_OK;
}

size_t im_dev_flushgroup_scratch_bytes(int32_t n_slots_cap, int32_t n_flushes_cap) { return im::flushgroup_scratch_bytes(n_slots_cap, n_flushes_cap); }

int im_dev_flushgroup_scratch_init(im_ctx* ctx, int32_t n_slots_cap, int32_t n_flushes_cap, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || !scratch || n_slots_cap < 0 || n_flushes_cap < 0 || scratch_bytes < im::flushgroup_scratch_bytes(n_slots_cap, n_flushes_cap)) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_flushgroup_init(n_slots_cap, n_flushes_cap, scratch, (hipStream_t)stream));
    { std::lock_guard<std::mutex> lk(ctx->gb_mu); ctx->fg_layout[scratch] = std::make_pair(n_slots_cap, n_flushes_cap); }
    return IM_OK;
}

int im_dev_flush_groupby(im_ctx* ctx, const im_flush_desc* desc_dev, int32_t n_flushes,
                         const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                         const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base, int32_t pe_count, int32_t tie_desc,
                         int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                         void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || n_flushes < 0 || !desc_dev || !cand_rec || !n_cand_dev || cand_cap < 0 || pe_count < 0 || !counts || !scratch) return IM_E_ARG;
    if (((((uintptr_t)cls) | ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)consumed) | ((uintptr_t)desc_dev) | ((uintptr_t)cl_key)) & 15u) || (((uintptr_t)counts) & 7u)) {
        set_err(ctx, "im_dev_flush_groupby: the slot arrays, desc and cl_key must be 16-byte aligned, counts 8-byte aligned"); return IM_E_ARG;
    }
    std::pair<int32_t, int32_t> lay(-1, -1);
    {
        std::lock_guard<std::mutex> lk(ctx->gb_mu);
        auto it = ctx->fg_layout.find(scratch);
        if (it != ctx->fg_layout.end()) lay = it->second;
    }
    if (lay.first < 0) { set_err(ctx, "flush + group-by scratch was not initialised (im_dev_flushgroup_scratch_init)"); return IM_E_ARG; }
    if ((int64_t)cand_cap * IM_MAX_EV > lay.first || n_flushes > lay.second || scratch_bytes < im::flushgroup_scratch_bytes(lay.first, lay.second)) {
        set_err(ctx, "flush + group-by scratch too small"); return IM_E_ARG;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_flush_groupby(lay.first, lay.second, desc_dev, n_flushes, cls, b1, b2, consumed, cand_rec, n_cand_dev, cand_cap, pe_base, pe_count,
                                          tie_desc, order, cl_key, cl_first, cl_count, counts, scratch, (hipStream_t)stream));
    return IM_OK;
}

size_t im_dev_groupby_scratch_bytes(int32_t n_slots) { return im::groupby_scratch_bytes(n_slots); }

int im_dev_groupby_scratch_init(im_ctx* ctx, int32_t n_slots, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || !scratch || scratch_bytes < im::groupby_scratch_bytes(n_slots)) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_groupby_init(n_slots, scratch, (hipStream_t)stream));
    { std::lock_guard<std::mutex> lk(ctx->gb_mu); ctx->gb_layout[scratch] = n_slots; }
    return IM_OK;
}

// the slot count a group-by scratch was initialised for, or -1
static int32_t groupby_layout(im_ctx* ctx, void* scratch)
{
    std::lock_guard<std::mutex> lk(ctx->gb_mu);
    auto it = ctx->gb_layout.find(scratch);
    return it == ctx->gb_layout.end() ? -1 : it->second;
}

int im_dev_cluster_groupby(im_ctx* ctx, int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                           const int32_t* consumed, int32_t tie_desc,
                           int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                           void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || n_slots < 0 || !counts) return IM_E_ARG;
    const int32_t n_layout = groupby_layout(ctx, scratch);
    if (n_layout < 0) { set_err(ctx, "group-by scratch was not initialised (im_dev_groupby_scratch_init)"); return IM_E_ARG; }
    if (n_slots > n_layout || scratch_bytes < im::groupby_scratch_bytes(n_layout)) { set_err(ctx, "group-by scratch too small"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_slots == 0) { HIP_TRY(ctx, hipMemsetAsync(counts, 0, 8, (hipStream_t)stream)); return IM_OK; }
    HIP_TRY(ctx, im::launch_groupby(n_layout, n_slots, nullptr, cls, b1, b2, consumed, tie_desc, order, cl_key, cl_first, cl_count, counts, scratch, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_cluster_groupby_n(im_ctx* ctx, int32_t n_slots_cap, const int32_t* n_cand_dev, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                             const int32_t* consumed, int32_t tie_desc,
                             int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                             void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx || n_slots_cap <= 0 || !counts || !n_cand_dev) return IM_E_ARG;
    const int32_t n_layout = groupby_layout(ctx, scratch);
    if (n_layout < 0) { set_err(ctx, "group-by scratch was not initialised (im_dev_groupby_scratch_init)"); return IM_E_ARG; }
    if (n_slots_cap > n_layout || scratch_bytes < im::groupby_scratch_bytes(n_layout)) { set_err(ctx, "group-by scratch too small"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_groupby(n_layout, n_slots_cap, n_cand_dev, cls, b1, b2, consumed, tie_desc, order, cl_key, cl_first, cl_count, counts, scratch, (hipStream_t)stream));
    return IM_OK;
}

// ---- seam 3, genome-wide form ------------------------------------------------------------------

int im_depth_enable(im_ctx* ctx)
{
    if (!ctx) return IM_E_ARG;
    if (!ctx->ref_ascii) { set_err(ctx, "im_set_reference has not been called"); return IM_E_ARG; }
    if (ctx->gdepth) return IM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int64_t tiles = 0;
    ctx->h_sums_off.clear();
    for (int32_t l : ctx->h_len) { ctx->h_sums_off.push_back(tiles); tiles += im::depth_sums_ints(l); }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->gdepth, (size_t)ctx->ref_total * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->gdepth_sums, (size_t)(tiles + 1) * sizeof(int32_t)));
    HIP_TRY(ctx, hipMemsetAsync(ctx->gdepth, 0, (size_t)ctx->ref_total * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->gdepth_sums, 0, (size_t)(tiles + 1) * sizeof(int32_t), ctx->stream));      // the arrival counters start at zero
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IM_OK;
}

int im_depth_scan(im_ctx* ctx, int32_t tid, void* stream)
{
    if (!ctx || !ctx->gdepth || tid < 0 || tid >= ctx->n_contigs) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one launch: tile-local depths + exclusive tile offsets (im_depth_query_tid adds them)
    HIP_TRY(ctx, im::launch_depth_scan_tiled(ctx->gdepth + ctx->h_asc_off[tid], (int64_t)ctx->h_len[tid] + 1, ctx->gdepth_sums + ctx->h_sums_off[tid], (hipStream_t)stream));
    return IM_OK;
}

int im_depth_reset(im_ctx* ctx, int32_t tid, void* stream)
{
    if (!ctx || !ctx->gdepth || tid < 0 || tid >= ctx->n_contigs) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->gdepth + ctx->h_asc_off[tid], 0, ((size_t)ctx->h_len[tid] + 1) * sizeof(int32_t), (hipStream_t)stream));
    return IM_OK;
}

int im_depth_allreduce(im_ctx* ctx, im_comm* comm)
{
    if (!ctx || !comm) return IM_E_ARG;
    if (!ctx->gdepth) { set_err(ctx, "im_depth_enable has not been called"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t total = (size_t)ctx->ref_total, step = (size_t)1 << 28;           // 1 GiB of int32 per call
    for (size_t at = 0; at < total; at += step) {
        const size_t n = total - at < step ? total - at : step;
        if (im_comm_allreduce_sum_i32(comm, ctx->gdepth + at, n, ctx->stream) != IM_OK) { set_err(ctx, "%s", im_comm_last_error()); return IM_E_HIP; }
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IM_OK;
}

int im_depth_query_max_tid(im_ctx* ctx, int32_t tid, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out, uint32_t* max_out)
{
    if (!ctx || n < 0 || !ctx->gdepth || tid < 0 || tid >= ctx->n_contigs) return IM_E_ARG;
    if (n == 0) return IM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sb = up256(sizeof(int32_t) * (size_t)n);
    int rc = ensure_ws(ctx, 4 * sb);
    if (rc) return rc;
    int32_t* d_beg = (int32_t*)ctx->ws;
    int32_t* d_end = (int32_t*)((char*)ctx->ws + sb);
    uint32_t* d_out = (uint32_t*)((char*)ctx->ws + 2 * sb);
    uint32_t* d_max = max_out ? (uint32_t*)((char*)ctx->ws + 3 * sb) : nullptr;
    HIP_TRY(ctx, hipMemcpyAsync(d_beg, beg, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_end, end, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, im::launch_depth_query_tiled(n, d_beg, d_end, ctx->gdepth + ctx->h_asc_off[tid], ctx->gdepth_sums + ctx->h_sums_off[tid], ctx->h_len[tid], d_out, d_max, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(sum_out, d_out, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (max_out) HIP_TRY(ctx, hipMemcpyAsync(max_out, d_max, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IM_OK;
}

int im_depth_query_tid(im_ctx* ctx, int32_t tid, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out)
{
    return im_depth_query_max_tid(ctx, tid, n, beg, end, sum_out, nullptr);
}

// Host-buffer entry point.  The batch is cut into chunks that travel through a two-slot pipeline: chunk c is
// packed into PINNED staging memory and copied in on one copy stream while chunk c-1 runs on the compute stream
// and the results of chunk c-2 come back on another (hipMemcpyAsync from / to pinned memory throughout, stream-to-stream
// events between the three stages), so that PCIe, the kernel and the host-side packing overlap.
int im_realign_batch(im_ctx* ctx, const im_params* params, const im_read_batch* batch, im_read_result* out)
{
    if (!ctx || !batch || !out) return IM_E_ARG;
    int rc = check_params(ctx, params);
    if (rc) return rc;
    const int32_t n = batch->n;
    if (n < 0) { set_err(ctx, "negative batch size"); return IM_E_ARG; }
    if (n == 0) return IM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->back_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_in[i], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_k[i], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_out[i], hipEventDisableTiming));
        }
    }
    for (int32_t i = 0; i < n; i++) {
        const int64_t l = batch->base_off[i + 1] - batch->base_off[i];
        if (l < 0 || l > 0x7fffffff) { set_err(ctx, "read %d has a bad length", i); return IM_E_ARG; }
    }

    constexpr int32_t kChunk = 32768;                      // reads per pipeline stage
    const int32_t cn = n < kChunk ? n : kChunk;
    // per-slot layout (identical on the device and in pinned memory): bases (worst case from the batch), off, len, tid, anchor, range, results
    int64_t max_bases = 0;
    for (int32_t c0 = 0; c0 < n; c0 += kChunk) {
        const int32_t c1 = c0 + kChunk < n ? c0 + kChunk : n;
        int64_t b = 0;
        for (int32_t i = c0; i < c1; i++) b += ((batch->base_off[i + 1] - batch->base_off[i]) + 3) & ~(int64_t)3;
        if (b > max_bases) max_bases = b;
    }
    const size_t bases_bytes = up256((size_t)max_bases + 16);
    const size_t off_bytes = up256(sizeof(int64_t) * (size_t)cn);
    const size_t i32_bytes = up256(sizeof(int32_t) * (size_t)cn);
    const size_t res_bytes = up256(sizeof(im_read_result) * (size_t)cn);
    const size_t in_bytes = bases_bytes + off_bytes + 4 * i32_bytes;
    const size_t slot_bytes = in_bytes + res_bytes;
    rc = ensure_ws(ctx, 2 * slot_bytes);
    if (rc) return rc;
    rc = ensure_pin(ctx, 2 * slot_bytes);
    if (rc) return rc;

    const int32_t nchunks = (n + kChunk - 1) / kChunk;
    auto collect = [&](int32_t c) -> int {                 // results of chunk c: pinned -> caller
        const int sl = c & 1;
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_out[sl]));
        const int32_t c0 = c * kChunk, c1 = c0 + kChunk < n ? c0 + kChunk : n;
        memcpy(out + c0, (char*)ctx->pin + (size_t)sl * slot_bytes + in_bytes, sizeof(im_read_result) * (size_t)(c1 - c0));
        return IM_OK;
    };
    for (int32_t c = 0; c < nchunks; c++) {
        const int sl = c & 1;
        if (c >= 2) { rc = collect(c - 2); if (rc) return rc; }      // frees this slot's pinned and device halves
        const int32_t c0 = c * kChunk, c1 = c0 + kChunk < n ? c0 + kChunk : n, m = c1 - c0;
        char* hp = (char*)ctx->pin + (size_t)sl * slot_bytes;
        char* dp = (char*)ctx->ws + (size_t)sl * slot_bytes;
        uint8_t* h_bases = (uint8_t*)hp;
        int64_t* h_off = (int64_t*)(hp + bases_bytes);
        int32_t* h_len = (int32_t*)(hp + bases_bytes + off_bytes);
        int32_t* h_tid = (int32_t*)(hp + bases_bytes + off_bytes + i32_bytes);
        int32_t* h_anchor = (int32_t*)(hp + bases_bytes + off_bytes + 2 * i32_bytes);
        int32_t* h_range = (int32_t*)(hp + bases_bytes + off_bytes + 3 * i32_bytes);
        // re-pack the reads at 4-byte aligned offsets (device layout requirement), straight into pinned memory
        int64_t pos = 0, longest = 0;
        for (int32_t i = 0; i < m; i++) {
            const int64_t l = batch->base_off[c0 + i + 1] - batch->base_off[c0 + i];
            h_off[i] = pos; h_len[i] = (int32_t)l;
            if (l > longest) longest = l;
            memcpy(h_bases + pos, batch->bases + batch->base_off[c0 + i], (size_t)l);
            const int64_t padded = (l + 3) & ~(int64_t)3;
            memset(h_bases + pos + l, 0, (size_t)(padded - l));
            pos += padded;
        }
        memset(h_bases + pos, 0, 16);
        memcpy(h_tid, batch->tid + c0, sizeof(int32_t) * (size_t)m);
        memcpy(h_anchor, batch->anchor + c0, sizeof(int32_t) * (size_t)m);
        memcpy(h_range, batch->range_max + c0, sizeof(int32_t) * (size_t)m);
        if (longest > im::kShortRead) im_expect_read_length(ctx, (int32_t)longest);
        HIP_TRY(ctx, hipMemcpyAsync(dp, hp, in_bytes, hipMemcpyHostToDevice, ctx->copy_stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_in[sl], ctx->copy_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_in[sl], 0));

        im_dev_batch db;
        db.n = m; db.bases = (uint8_t*)dp; db.base_off = (int64_t*)(dp + bases_bytes);
        db.read_len = (int32_t*)(dp + bases_bytes + off_bytes); db.tid = (int32_t*)(dp + bases_bytes + off_bytes + i32_bytes);
        db.anchor = (int32_t*)(dp + bases_bytes + off_bytes + 2 * i32_bytes); db.range_max = (int32_t*)(dp + bases_bytes + off_bytes + 3 * i32_bytes);
        db.out = (im_read_result*)(dp + in_bytes);
        db.ev_cls = nullptr; db.ev_b1 = nullptr; db.ev_b2 = nullptr;
        rc = im_dev_realign(ctx, params, &db, ctx->stream);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k[sl], ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->back_stream, ctx->ev_k[sl], 0));
        HIP_TRY(ctx, hipMemcpyAsync(hp + in_bytes, dp + in_bytes, sizeof(im_read_result) * (size_t)m, hipMemcpyDeviceToHost, ctx->back_stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_out[sl], ctx->back_stream));
    }
    for (int32_t c = nchunks >= 2 ? nchunks - 2 : 0; c < nchunks; c++) { rc = collect(c); if (rc) return rc; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

    int worst = IM_OK;
    for (int32_t i = 0; i < n; i++) {
        const int st = out[i].status;
        if (st == IM_ST_ABORT && worst == IM_OK) { worst = IM_E_ABORT; set_err(ctx, "read %d: the reference would abort on this input", i); }
        else if (st == IM_ST_OVERFLOW && worst == IM_OK) { worst = IM_E_OVERFLOW; set_err(ctx, "read %d: segment list longer than IM_MAX_OPS", i); }
        else if (st == IM_ST_UNSUPPORTED && worst == IM_OK) { worst = IM_E_UNSUPPORTED; set_err(ctx, "read %d: longer than IM_MAX_READ=%d%s", i, IM_MAX_READ, params->numgaps ? ", or than 255 with numgaps > 0" : ""); }
    }
    return worst;
}

int im_depth_build(im_ctx* ctx, int64_t contig_len, int32_t n_seg, const int32_t* seg_start, const int32_t* seg_len)
{
    if (!ctx || contig_len < 0 || n_seg < 0) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (contig_len + 1 > ctx->depth_cap) {
        if (ctx->depth) { HIP_TRY(ctx, hipFree(ctx->depth)); ctx->depth = nullptr; }
        if (ctx->depth_sums) { HIP_TRY(ctx, hipFree(ctx->depth_sums)); ctx->depth_sums = nullptr; }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->depth, (size_t)(contig_len + 1) * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->depth_sums, (size_t)(im::depth_tiles(contig_len) + 1) * sizeof(int32_t)));
        ctx->depth_cap = contig_len + 1;
    }
    const size_t sb = up256(sizeof(int32_t) * (size_t)(n_seg ? n_seg : 1));
    int rc = ensure_ws(ctx, 2 * sb);
    if (rc) return rc;
    int32_t* d_start = (int32_t*)ctx->ws;
    int32_t* d_len = (int32_t*)((char*)ctx->ws + sb);
    if (n_seg > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_start, seg_start, sizeof(int32_t) * (size_t)n_seg, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_len, seg_len, sizeof(int32_t) * (size_t)n_seg, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, im::launch_depth_build(contig_len, n_seg, d_start, d_len, ctx->depth, ctx->depth_sums, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->depth_len = contig_len;
    return IM_OK;
}

int im_depth_query(im_ctx* ctx, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out)
{
    if (!ctx || n < 0) return IM_E_ARG;
    if (ctx->depth_len < 0) { set_err(ctx, "im_depth_build has not been called"); return IM_E_ARG; }
    if (n == 0) return IM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sb = up256(sizeof(int32_t) * (size_t)n);
    int rc = ensure_ws(ctx, 3 * sb);
    if (rc) return rc;
    int32_t* d_beg = (int32_t*)ctx->ws;
    int32_t* d_end = (int32_t*)((char*)ctx->ws + sb);
    uint32_t* d_out = (uint32_t*)((char*)ctx->ws + 2 * sb);
    HIP_TRY(ctx, hipMemcpyAsync(d_beg, beg, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_end, end, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, im::launch_depth_query(n, d_beg, d_end, ctx->depth, ctx->depth_len, d_out, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(sum_out, d_out, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IM_OK;
}

int im_support_batch(im_ctx* ctx, int32_t n, const uint8_t* targets, const int64_t* t_off,
                     const uint8_t* queries, const int64_t* q_off, int32_t* out)
{
    if (!ctx || n < 0) return IM_E_ARG;
    if (n == 0) return IM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t tb = up256((size_t)t_off[n] + 16), qb = up256((size_t)q_off[n] + 16);
    const size_t ob = up256(sizeof(int64_t) * ((size_t)n + 1)), rb = up256(sizeof(int32_t) * 4 * (size_t)n);
    int64_t max_t = 0, max_q = 0;                                // max_t sizes the kernel's LDS; beyond the LDS form's bounds: the second launch
    for (int32_t i = 0; i < n; i++) {
        const int64_t l = t_off[i + 1] - t_off[i], q = q_off[i + 1] - q_off[i];
        if (l > max_t) max_t = l;
        if (q > max_q) max_q = q;
    }
    int32_t big_grid = 0;
    const size_t bigb = up256(im::support_big_scratch_bytes(max_t, max_q, n, &big_grid));
    int rc = ensure_ws(ctx, tb + qb + 2 * ob + rb + bigb);
    if (rc) return rc;
    char* w = static_cast<char*>(ctx->ws);
    uint8_t* d_t = (uint8_t*)w; w += tb;
    uint8_t* d_q = (uint8_t*)w; w += qb;
    int64_t* d_to = (int64_t*)w; w += ob;
    int64_t* d_qo = (int64_t*)w; w += ob;
    int32_t* d_out = (int32_t*)w; w += rb;
    void* d_big = (void*)w;
    HIP_TRY(ctx, hipMemcpyAsync(d_t, targets, (size_t)t_off[n], hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_q, queries, (size_t)q_off[n], hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_to, t_off, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_qo, q_off, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, im::launch_support(n, d_t, d_to, d_q, d_qo, d_out, max_t, max_q, d_big, big_grid, ctx->n_cu, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out, d_out, sizeof(int32_t) * 4 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int32_t i = 0; i < n; i++)
        if (out[4 * i + 3] == IM_ST_UNSUPPORTED) { set_err(ctx, "support task %d: target or query longer than the kernel holds", i); return IM_E_UNSUPPORTED; }
    return IM_OK;
}

size_t im_dev_cluster_scratch_bytes(int32_t n) { return im::cluster_scratch_bytes(n); }
size_t im_dev_gather_scratch_bytes(int32_t n) { return im::gather_scratch_bytes(n); }

int im_dev_cluster_sr(im_ctx* ctx, int32_t n_cap, const int32_t* n_dev,
                      const int32_t* cls, const int32_t* b1, const int32_t* b2,
                      int32_t marker, int32_t tie_desc,
                      int32_t* order, int32_t* cl_first, int32_t* cl_count, uint8_t* used, int32_t* n_clusters,
                      void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    if (n_cap < 0 || !n_dev) { set_err(ctx, "bad evidence count arguments"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_cluster_sr(n_cap, n_dev, cls, b1, b2, marker, tie_desc, order, cl_first, cl_count, used, n_clusters,
                                       scratch, scratch_bytes, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_cluster_slots_max(void) { return im::cluster_small_max(); }

int im_dev_cluster_records(im_ctx* ctx, int32_t tid, const int32_t* counts,
                           const int32_t* order, const int32_t* cl_first, const int32_t* cl_count,
                           const int32_t* cls, const int32_t* b1, const int32_t* b2,
                           int32_t* recs, int32_t cap, void* stream)
{
    if (!ctx || cap < 1) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_cluster_records(tid, counts, order, cl_first, cl_count, cls, b1, b2, recs, cap, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_cluster_slots(im_ctx* ctx, int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                         int32_t marker, int32_t tie_desc,
                         int32_t* order, int32_t* cl_first, int32_t* cl_count, uint8_t* used, int32_t* counts, void* stream)
{
    if (!ctx) return IM_E_ARG;
    if (n_slots < 0 || !counts) { set_err(ctx, "bad slot arguments"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_cluster_small(n_slots, nullptr, cls, b1, b2, marker, tie_desc, order, cl_first, cl_count,
                                          used, counts, (hipStream_t)stream));
    return IM_OK;
}

size_t im_dev_cluster_hist_scratch_bytes(int32_t n_slots) { return im::cluster_hist_scratch_bytes(n_slots); }

int im_dev_cluster_hist_init(im_ctx* ctx, int32_t n_slots, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_cluster_hist_init(n_slots, scratch, scratch_bytes, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_cluster_hist(im_ctx* ctx, int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                        int32_t marker, int32_t tie_desc,
                        int32_t* order, int32_t* cl_first, int32_t* cl_count, uint8_t* used, int32_t* counts,
                        void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    if (n_slots < 0 || !counts) { set_err(ctx, "bad slot arguments"); return IM_E_ARG; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_cluster_hist(n_slots, cls, b1, b2, marker, tie_desc, order, cl_first, cl_count, used, counts,
                                         scratch, scratch_bytes, (hipStream_t)stream));
    return IM_OK;
}

int im_dev_gather_evidence(im_ctx* ctx, const im_read_result* res, int32_t n,
                           int32_t* cls, int32_t* b1, int32_t* b2, int32_t* src,
                           int32_t cap, int32_t* n_out, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, im::launch_gather_evidence(res, n, cls, b1, b2, src, cap, n_out, scratch, scratch_bytes, (hipStream_t)stream));
    return IM_OK;
}

int im_cluster_sr(im_ctx* ctx, int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                  int32_t marker, int32_t tie_desc,
                  int32_t* order, int32_t* cl_first, int32_t* cl_count, uint8_t* used, int32_t* n_clusters)
{
    if (!ctx || !n_clusters) return IM_E_ARG;
    if (n < 0) { set_err(ctx, "negative evidence count"); return IM_E_ARG; }
    *n_clusters = 0;
    if (n == 0) return IM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t a32 = up256(sizeof(int32_t) * (size_t)n);
    const size_t scratch = im::cluster_scratch_bytes(n);
    const size_t hist_bytes = im::cluster_hist_scratch_bytes(n);
    int rc = ensure_ws(ctx, 6 * a32 + up256((size_t)n) + 256 + up256(scratch) + hist_bytes);
    if (rc) return rc;
    char* w = static_cast<char*>(ctx->ws);
    int32_t* d_cls = (int32_t*)w; w += a32;
    int32_t* d_b1 = (int32_t*)w; w += a32;
    int32_t* d_b2 = (int32_t*)w; w += a32;
    int32_t* d_order = (int32_t*)w; w += a32;
    int32_t* d_first = (int32_t*)w; w += a32;
    int32_t* d_count = (int32_t*)w; w += a32;
    uint8_t* d_used = (uint8_t*)w; w += up256((size_t)n);
    int32_t* d_ncl = (int32_t*)w; w += 128;
    int32_t* d_n = (int32_t*)w; w += 128;
    void* d_scratch = w; w += up256(scratch);
    void* d_hist = w;
    HIP_TRY(ctx, hipMemcpyAsync(d_cls, cls, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_b1, b1, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_b2, b2, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_n, &n, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    {
        // breakpoint-histogram path: every record is a live slot
        HIP_TRY(ctx, im::launch_cluster_hist_init(n, d_hist, hist_bytes, ctx->stream));
        HIP_TRY(ctx, im::launch_cluster_hist(n, d_cls, d_b1, d_b2, marker, tie_desc, d_order, d_first, d_count, d_used, d_ncl,
                                             d_hist, hist_bytes, ctx->stream));
        int32_t got = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&got, d_ncl, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (got < 0) {      // more distinct breakpoints / deeper clusters than the histogram holds: radix path
            rc = im_dev_cluster_sr(ctx, n, d_n, d_cls, d_b1, d_b2, marker, tie_desc, d_order, d_first, d_count, d_used, d_ncl,
                                   d_scratch, scratch, ctx->stream);
            if (rc) return rc;
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(n_clusters, d_ncl, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int32_t ncl = *n_clusters;
    HIP_TRY(ctx, hipMemcpyAsync(order, d_order, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(used, d_used, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (ncl > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(cl_first, d_first, sizeof(int32_t) * (size_t)ncl, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(cl_count, d_count, sizeof(int32_t) * (size_t)ncl, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IM_OK;
}

struct im_timer { im_ctx* ctx; hipEvent_t a, b; };

int im_dev_alloc(im_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(out, bytes ? bytes : 1));
    return IM_OK;
}
int im_dev_free(im_ctx* ctx, void* p)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { std::lock_guard<std::mutex> lk(ctx->gb_mu); ctx->gb_layout.erase(p); ctx->fg_layout.erase(p); }     // a group-by scratch: its address may come back as something else
    HIP_TRY(ctx, hipFree(p));
    return IM_OK;
}
int im_dev_memset(im_ctx* ctx, void* dst_dev, int byte, size_t bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(dst_dev, byte, bytes, (hipStream_t)stream));
    return IM_OK;
}
int im_host_alloc(im_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return IM_OK;
}
int im_host_free(im_ctx* ctx, void* p)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipHostFree(p));
    return IM_OK;
}
int im_dev_upload_async(im_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return IM_OK;
}
int im_dev_download_async(im_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return IM_OK;
}
int im_dev_copy_async(im_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return IM_OK;
}
int im_dev_upload(im_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return IM_OK;
}
int im_dev_download(im_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return IM_OK;
}
void* im_ctx_stream(im_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int im_ctx_device(im_ctx* ctx) { return ctx ? ctx->device : -1; }
int im_stream_sync(im_ctx* ctx, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return IM_OK;
}
int im_timer_create(im_ctx* ctx, im_timer** out)
{
    if (!ctx || !out) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    im_timer* t = new im_timer();
    t->ctx = ctx;
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e != hipSuccess) { set_err(ctx, "hipEventCreate: %s", hipGetErrorString(e)); delete t; return IM_E_HIP; }
    *out = t;
    return IM_OK;
}
void im_timer_destroy(im_timer* t)
{
    if (!t) return;
    (void)hipEventDestroy(t->a); (void)hipEventDestroy(t->b);
    delete t;
}
int im_timer_start(im_timer* t, void* stream)
{
    if (!t) return IM_E_ARG;
    HIP_TRY(t->ctx, hipEventRecord(t->a, (hipStream_t)stream));
    return IM_OK;
}
int im_timer_stop(im_timer* t, void* stream)
{
    if (!t) return IM_E_ARG;
    HIP_TRY(t->ctx, hipEventRecord(t->b, (hipStream_t)stream));
    return IM_OK;
}
int im_timer_elapsed_ms(im_timer* t, float* ms)
{
    if (!t || !ms) return IM_E_ARG;
    HIP_TRY(t->ctx, hipEventSynchronize(t->b));
    HIP_TRY(t->ctx, hipEventElapsedTime(ms, t->a, t->b));
    return IM_OK;
}

// ---- extra streams and events: overlap of one flush's clustering with the next flush's realign ------
struct im_event { im_ctx* ctx; hipEvent_t e; };

int im_stream_create(im_ctx* ctx, void** out)
{
    if (!ctx || !out) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // highest priority: the work put here is a handful of small kernels that must find wave slots while a
    // chip-filling realign launch of the context's own stream is in flight
    int lo = 0, hi = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t st = nullptr;
    HIP_TRY(ctx, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
    *out = (void*)st;
    return IM_OK;
}
int im_stream_destroy(im_ctx* ctx, void* stream)
{
    if (!ctx || !stream) return IM_E_ARG;
    HIP_TRY(ctx, hipStreamDestroy((hipStream_t)stream));
    return IM_OK;
}
int im_event_create(im_ctx* ctx, im_event** out)
{
    if (!ctx || !out) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t e = nullptr;
    HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    im_event* r = new im_event();
    r->ctx = ctx; r->e = e;
    *out = r;
    return IM_OK;
}
void im_event_destroy(im_event* ev)
{
    if (!ev) return;
    (void)hipEventDestroy(ev->e);
    delete ev;
}
int im_event_record(im_event* ev, void* stream)
{
    if (!ev) return IM_E_ARG;
    HIP_TRY(ev->ctx, hipEventRecord(ev->e, (hipStream_t)stream));
    return IM_OK;
}
// record on `from`, make `to` wait: one call for the usual producer -> consumer hand-over
int im_stream_follow(im_event* ev, void* from, void* to)
{
    if (!ev) return IM_E_ARG;
    HIP_TRY(ev->ctx, hipEventRecord(ev->e, (hipStream_t)from));
    HIP_TRY(ev->ctx, hipStreamWaitEvent((hipStream_t)to, ev->e, 0));
    return IM_OK;
}
int im_event_sync(im_event* ev)
{
    if (!ev) return IM_E_ARG;
    HIP_TRY(ev->ctx, hipEventSynchronize(ev->e));
    return IM_OK;
}
int im_stream_wait_event(im_ctx* ctx, void* stream, im_event* ev)
{
    if (!ctx || !ev) return IM_E_ARG;
    HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)stream, ev->e, 0));
    return IM_OK;
}

// ---- launch graphs -----------------------------------------------------------------
// A flush is a fixed sequence of small dependent launches (realign, then the cluster kernels); captured
// once into a HIP graph it is replayed with one host call and without per-launch submission gaps.
struct im_graph { im_ctx* ctx; hipGraph_t g; hipGraphExec_t x; };

int im_capture_begin(im_ctx* ctx, void* stream)
{
    if (!ctx) return IM_E_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return IM_OK;
}
int im_capture_end(im_ctx* ctx, void* stream, im_graph** out)
{
    if (!ctx || !out) return IM_E_ARG;
    hipGraph_t g = nullptr;
    HIP_TRY(ctx, hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t x = nullptr;
    hipError_t e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); set_err(ctx, "hipGraphInstantiate: %s", hipGetErrorString(e)); return IM_E_HIP; }
    im_graph* r = new im_graph();
    r->ctx = ctx; r->g = g; r->x = x;
    *out = r;
    return IM_OK;
}
int im_graph_launch(im_graph* g, void* stream)
{
    if (!g) return IM_E_ARG;
    HIP_TRY(g->ctx, hipGraphLaunch(g->x, (hipStream_t)stream));
    return IM_OK;
}
void im_graph_destroy(im_graph* g)
{
    if (!g) return;
    (void)hipGraphExecDestroy(g->x); (void)hipGraphDestroy(g->g);
    delete g;
}

}  // extern "C"
