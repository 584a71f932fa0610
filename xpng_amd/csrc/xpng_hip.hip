// xpng_hip.hip -- C-ABI of libxpng_hip.so (include/xpng_hip.h): context, launches, host-buffer wrappers.
// gfx950 only.  No CPU fallback: every compute entry point fails when HIP has no device.
#include "../../include/xpng_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "m1_decode.hpp"
#include "normalize.hpp"
#include "m1_encode.hpp"
#include "m2_decode.hpp"
#include "m2_encode.hpp"
#include "rans2.hpp"
#include "rans2_wide.hpp"
#include "rans1_wide.hpp"
#include "rans1_wide_dec.hpp"
#include "tile_container.hpp"

using namespace xpng;

static thread_local std::string g_err;
static int fail(const std::string &m) { g_err = m; return 1; }
#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

// A kernel that dereferences a device pointer nobody allocated does not fail - it FAULTS: the ROCm runtime's fault handler prints
// "Memory access fault by GPU node ..." and calls abort(), past every catch of the extern "C" entry points (that is what took
// the test process down in gpurun_out/r2_t14.log: a working tree in which the five symbol planes had just become lazily
// allocated and launch_transform, reached through the mode-2 encode, did not yet call ensure_planes).  So every launch
// sequence names the workspace buffers it is about to hand to kernels and refuses to launch when one of them is null.
#define XPNG_REQUIRE(...)                                                                                                    \
    do {                                                                                                                     \
        const void *rq_[] = {__VA_ARGS__};                                                                                   \
        for (const void *q_ : rq_)                                                                                           \
            if (!q_) return fail("internal error: a workspace buffer of this launch sequence was never allocated (one of: " #__VA_ARGS__ ")"); \
    } while (0)

// A context forks a side stream and a pipelined caller keeps several contexts in flight: with the runtime's default of 4 hardware
// queues those streams share queues and serialise (measured: 31 -> 21 Gpx/s).  The runtime reads GPU_MAX_HW_QUEUES when it
// initialises - the first HIP call of the process - so the library asks for 32 when it is LOADED, unless the caller has chosen a
// value.  A process that has already initialised HIP before loading this library keeps what it had (export the variable yourself in
// that case: INTEGRATION.md).  Nothing in the library reads the variable back: no code path depends on the queues being there
// (round 3's decode took a third stream per context when it saw >= 24 here - ADVICE r3; the split needs no third stream any more).
__attribute__((constructor)) static void xpnghip_on_load(void) { (void)setenv("GPU_MAX_HW_QUEUES", "32", 0); }

extern "C" int xpnghip_abi_version(void) { return XPNGHIP_ABI_VERSION; }
extern "C" const char *xpnghip_last_error(void) { return g_err.c_str(); }
extern "C" int xpnghip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- tile table (reference compute_props_of_each_tile, libxpng.c:51-83) ---------------------------------
static void split_axis(uint64_t len, uint64_t base, uint64_t &count, uint64_t &first, uint64_t &second) {
    const uint64_t rem = len % base;
    count = len / base; first = base + rem; second = base;
    if (rem > base / 2) { count++; second = first / 2; first = second + (first & 1); }
}
// Tiles outside [r0, r1) get no plane / scratch space (a rank that encodes only its tile range of a large raster should
// not pay HBM for the rest); such a context can only be used on sub-ranges of [r0, r1).
static void build_tiles(uint64_t W, uint64_t H, std::vector<TileDesc> &out, uint64_t r0 = 0, uint64_t r1 = ~0ull) {
    uint64_t nx = 1, ny = 1, w0 = W, w1 = 0, bw = 0, h0 = H, h1 = 0, bh = 0;
    if (W * H > TILE_AREA) {
        if (W < 444) { bw = W; bh = TILE_AREA / W; }
        else if (H < 444) { bh = H; bw = TILE_AREA / H; }
        else bw = bh = 444;
        split_axis(W, bw, nx, w0, w1);
        split_axis(H, bh, ny, h0, h1);
    }
    out.clear();
    out.reserve(nx * ny);
    uint64_t pbase = 0, sbase = 0;
    for (uint64_t j = 0; j < ny; j++) {
        const uint64_t y = j == 0 ? 0 : (j == 1 ? h0 : h0 + h1 + (j - 2) * bh), th = j == 0 ? h0 : (j == 1 ? h1 : bh);
        for (uint64_t i = 0; i < nx; i++) {
            TileDesc t{};
            t.x = (uint32_t)(i == 0 ? 0 : (i == 1 ? w0 : w0 + w1 + (i - 2) * bw));
            t.w = (uint32_t)(i == 0 ? w0 : (i == 1 ? w1 : bw));
            t.y = (uint32_t)y; t.h = (uint32_t)th;
            t.n = t.w * t.h;
            t.pbase = pbase; t.sbase = sbase;
            const uint64_t idx = out.size();
            if (idx >= r0 && idx < r1) { pbase += rup(t.n + 192, 256); sbase += tile_scratch_bytes(t.n); }
            out.push_back(t);
        }
    }
}

struct xpnghip_ctx {
    int device = 0;
    uint64_t W = 0, H = 0;
    int pxsz = 0;
    uint32_t B = 1;    // images per launch (native batching: virtual tile = image * N + tile)
    uint64_t r0 = 0, r1 = 0;  // tile range this context has workspace for
    uint32_t spt = 0;  // streams per tile: 9 (+1 alpha)
    std::vector<TileDesc> tiles;  // the N tiles of ONE image (host copy); the device table has B * N entries
    uint64_t plane_img = 0, plane_stride = 0, scratch_img = 0, ws_bytes = 0;
    TileDesc *d_tiles = nullptr;
    uint8_t *d_planes = nullptr, *d_scratch = nullptr;  // d_planes (4 or 5 symbol planes) and d_scratch (level-1 stream scratch = the decode's planes): allocated on first use
    uint32_t *d_sums = nullptr, *d_nlh = nullptr, *d_ctx_n = nullptr, *d_k_n = nullptr, *d_blk_sz = nullptr, *d_tile_sz = nullptr, *d_tile_hdr = nullptr;
    uint64_t *d_off = nullptr, *d_totals = nullptr, *d_dbg = nullptr;
    uint64_t *d_blob_len = nullptr;       // decode: B blob lengths
    uint32_t *d_status = nullptr;         // decode: bit 0 = some tile failed header validation
    std::vector<uint64_t> h_blob_len;
    const uint8_t **d_in_ptrs = nullptr;  // B raster (encode) / blob (decode) pointers
    uint8_t **d_out_ptrs = nullptr;       // B blob (encode) / raster (decode) pointers
    std::vector<const void *> h_in_ptrs;  // what d_in_ptrs / d_out_ptrs currently hold (skip the upload when unchanged)
    std::vector<void *> h_out_ptrs;
    // decode keeps its own pair of tables: a caller that alternates encode and decode on one context (a pipeline) would
    // otherwise re-upload, and synchronise its stream, on every call
    uint32_t *d_order = nullptr;          // tile indices of [r0, r1) by decreasing pixel count (TileSel::order)
    // ONE side stream per context [r4]: the alpha chains of a batched encode run on it beside the context chains, and the alpha branch
    // of a decode (DecodeWs::side borrows it) - never at the same time, the calls on a context are ordered.  Every stream alive takes
    // a hardware queue, and the chip runs ~16-20 of them side by side before the pipeline slots start to cost each other (r04_experiments)
    hipStream_t enc_side = nullptr;
    hipEvent_t ev_enc_fork = nullptr, ev_enc_join = nullptr;
    const uint8_t **d_dec_in_ptrs = nullptr;
    uint8_t **d_dec_out_ptrs = nullptr;
    std::vector<const void *> h_dec_in_ptrs;
    std::vector<void *> h_dec_out_ptrs;
    WPrep *d_wprep = nullptr;   // wide entropy stage: per (tile, stream) record, encoder tables, normalised frequencies
    uint8_t *d_wtab = nullptr;
    uint32_t nlh_slots = 0, nlh_generic = 0;  // records per tile in d_nlh (m1_encode.hpp); which transform form wrote them last
    uint32_t n_big = 0, n_top = 0;  // tiles of the biggest size class (>= 3/4 of the largest pixel count); tiles as large as the largest
    hipStream_t enc_side2 = nullptr;      // the alpha streams of the biggest tiles, a wavefront each, beside the wide chains of the rest
    hipEvent_t ev_enc_join2 = nullptr;
    uint16_t *d_wF = nullptr;
    // mode 2 (RGB slow level): allocated on first use
    uint8_t *d_scratch2 = nullptr;
    uint64_t *d_sbase2 = nullptr;
    uint32_t *d_flags2 = nullptr, *d_stream_n2 = nullptr;
    M2Blk *d_blk2 = nullptr;
    M2Tile *d_mt2 = nullptr;
    M2DecTile *d_info2 = nullptr;
    uint16_t *d_tabs2 = nullptr;
    W1Prep *d_w1prep = nullptr;           // wide rANS v1 encode (rans1_wide.hpp): descriptors, encoder tables, frequencies
    uint8_t *d_w1tab = nullptr;
    uint16_t *d_w1F = nullptr;
    int stamps = 0;  // XPNG_STAMPS=1: chain kernels record s_memtime phase stamps (debug_fetch 40/41)
    uint64_t *h_total = nullptr;  // pinned, B entries
    hipStream_t stream = nullptr;
    // host-buffer wrappers keep their own device raster / blob buffers here (capacities in bytes; grown on demand, never per call)
    uint8_t *d_raster = nullptr, *d_blobs = nullptr, *d_blob_in = nullptr;
    uint64_t cap_raster = 0, cap_blobs = 0, cap_blob_in = 0;
    uint8_t *h_stage = nullptr;  // pinned host staging of a decoded band (wrappers.hpp: device -> pinned at link speed, pinned -> caller's raster by copy threads)
    uint64_t cap_stage = 0;
    uint64_t call = 0;  // id of the last host-wrapper call that used this context (wrappers.hpp: never evicted mid-call)
    DecodeWs dec;
};

// the context's own stream, created when a call first needs it (the `stream == NULL` form of the device-resident entry points,
// and the host-buffer wrappers).  On failure nullptr, i.e. the legacy default stream: slower, still correct.
static hipStream_t ctx_stream(xpnghip_ctx *c) {
    if (!c->stream) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (hipSetDevice(c->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) c->stream = nullptr;
        if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    }
    return c->stream;
}

extern "C" void xpnghip_ctx_destroy(xpnghip_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->d_tiles, c->d_planes, c->d_scratch, c->d_sums, c->d_nlh, c->d_ctx_n, c->d_k_n, c->d_blk_sz, c->d_tile_sz,
                    c->d_tile_hdr, c->d_off, c->d_totals, c->d_raster, c->d_blobs, c->d_blob_in, c->d_dbg, (void *)c->d_in_ptrs, (void *)c->d_out_ptrs, (void *)c->d_dec_in_ptrs, (void *)c->d_dec_out_ptrs, (void *)c->d_order,
                    c->d_wprep, c->d_wtab, c->d_wF, c->d_blob_len, c->d_status, c->d_scratch2, c->d_sbase2, c->d_flags2, c->d_stream_n2, c->d_blk2, c->d_mt2, c->d_info2, c->d_tabs2, c->d_w1prep, c->d_w1tab, c->d_w1F};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->enc_side) (void)hipStreamDestroy(c->enc_side);
    if (c->enc_side2) (void)hipStreamDestroy(c->enc_side2);
    if (c->ev_enc_join2) (void)hipEventDestroy(c->ev_enc_join2);
    if (c->ev_enc_fork) (void)hipEventDestroy(c->ev_enc_fork);
    if (c->ev_enc_join) (void)hipEventDestroy(c->ev_enc_join);
    decode_ws_free(c->dec);
    if (c->h_total) (void)hipHostFree(c->h_total);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int ctx_create_range_impl(xpnghip_ctx **out, int device, uint64_t w, uint64_t h, int pxsz, uint32_t batch, uint64_t r0, uint64_t r1);
extern "C" int xpnghip_ctx_create_range(xpnghip_ctx **out, int device, uint64_t w, uint64_t h, int pxsz, uint32_t batch,
                                        uint64_t r0, uint64_t r1) {
    try { return ctx_create_range_impl(out, device, w, h, pxsz, batch, r0, r1); }
    catch (const std::bad_alloc &) { return fail("out of host memory"); }
    catch (...) { return fail("unexpected C++ exception"); }
}
static int ctx_create_range_impl(xpnghip_ctx **out, int device, uint64_t w, uint64_t h, int pxsz, uint32_t batch, uint64_t r0, uint64_t r1) {
    if (!out || !w || !h || w > (1u << 24) || h > (1u << 24) || (pxsz != 3 && pxsz != 4) || batch < 1 || batch > 4096 || r0 >= r1) return fail("bad arguments");
    if (xpnghip_device_count() <= device || device < 0) return fail("no such HIP device (libxpng_hip has no CPU fallback)");
    HIPCHK(hipSetDevice(device));
    xpnghip_ctx *c = new xpnghip_ctx();
    c->device = device; c->W = w; c->H = h; c->pxsz = pxsz; c->spt = pxsz == 4 ? 10 : 9; c->B = batch;
    build_tiles(w, h, c->tiles, r0, r1);
    const uint64_t N = c->tiles.size(), VN = N * batch;
    c->r0 = r0; c->r1 = r1 < N ? r1 : N;
    const TileDesc &last = c->tiles[c->r1 - 1];
    c->plane_img = last.pbase + rup(last.n + 192, 256);
    c->scratch_img = last.sbase + tile_scratch_bytes(last.n);
    c->plane_stride = c->plane_img * batch;
    std::vector<TileDesc> all(VN);
    for (uint32_t b = 0; b < batch; b++)
        for (uint64_t i = 0; i < N; i++) {
            TileDesc t = c->tiles[i];
            t.img = b; t.pbase += b * c->plane_img; t.sbase += b * c->scratch_img;
            all[b * N + i] = t;
        }
#define ALLOC(ptr, bytes)                                                                          \
    do {                                                                                           \
        if (hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) {                                   \
            xpnghip_ctx_destroy(c);                                                                \
            return fail("hipMalloc failed for " #ptr);                                             \
        }                                                                                          \
        c->ws_bytes += (bytes);                                                                    \
    } while (0)
    ALLOC(c->d_tiles, VN * sizeof(TileDesc));
    // (the stream scratch - 7.5 B/px for a level-1 encode, 8 B/px as the decode's symbol / residual planes: one buffer of the larger
    //  size - and the symbol planes, 4 or 5 B/px, are allocated by the first call that needs them: ensure_scratch, ensure_planes)
    ALLOC(c->d_sums, VN * 16);
    for (uint64_t i = c->r0; i < c->r1; i++) {
        const TileDesc &t = c->tiles[i];
        c->nlh_slots = std::max(c->nlh_slots, std::max(nlh_records(t.w, t.h, false), nlh_records(t.w, t.h, true)));
    }
    ALLOC(c->d_nlh, VN * c->nlh_slots * NLH_STRIDE * 4 + 64);  // nl histogram + last coded pixel of every transform workgroup (m1_encode.hpp)
    ALLOC(c->d_ctx_n, VN * 9 * 4);
    ALLOC(c->d_k_n, VN * 4);
    ALLOC(c->d_blk_sz, VN * 10 * 4);
    ALLOC(c->d_tile_sz, VN * 4);
    ALLOC(c->d_tile_hdr, VN * 4);
    ALLOC(c->d_off, (VN + batch) * 8);
    ALLOC(c->d_totals, (uint64_t)batch * 8);
#ifdef XPNG_PROBES
    ALLOC(c->d_dbg, VN * 10 * 8 * 8 * 2);  // phase stamps of the chain kernels (XPNG_STAMPS)
#endif
    ALLOC(c->d_wprep, VN * 10 * sizeof(WPrep));
    ALLOC(c->d_wtab, VN * WTAB_TILE_BYTES + 4096 + watab_bytes(VN));  // per-tile tables, then the launch's interleaved alpha tables (rans2_wide.hpp)
    ALLOC(c->d_wF, VN * 10 * 512);
    ALLOC(c->d_blob_len, (uint64_t)batch * 8);
    ALLOC(c->d_status, 64);
    ALLOC(c->d_in_ptrs, (uint64_t)batch * 8);
    ALLOC(c->d_out_ptrs, (uint64_t)batch * 8);
    ALLOC(c->d_dec_in_ptrs, (uint64_t)batch * 8);
    ALLOC(c->d_dec_out_ptrs, (uint64_t)batch * 8);
    ALLOC(c->d_order, (c->r1 - c->r0) * 4);
#undef ALLOC
    {   // tiles of [r0, r1) by decreasing pixel count (stable: equal sizes stay in tile order)
        std::vector<uint32_t> ord;
        for (uint64_t i = c->r0; i < c->r1; i++) ord.push_back((uint32_t)i);
        std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return c->tiles[a].n > c->tiles[b].n; });
        if (hipMemcpy(c->d_order, ord.data(), ord.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { xpnghip_ctx_destroy(c); return fail("context setup failed"); }
        // size class of the biggest tiles (>= 3/4 of the largest pixel count): the decode walks them beside the rest (m1_decode.hpp)
        c->n_big = 0; c->n_top = 0;
        for (uint32_t i : ord) { if ((uint64_t)c->tiles[i].n * 4 >= (uint64_t)c->tiles[ord[0]].n * 3) c->n_big++; if (c->tiles[i].n == c->tiles[ord[0]].n) c->n_top++; }
    }
    c->stamps = c->d_dbg && probe_env("XPNG_STAMPS") != nullptr;  // (probe builds only)
#ifdef XPNG_PROBES
    {   // VALU burner (common.hpp)
        const uint32_t btr = (uint32_t)probe_pad("XPNG_BURN_TR"), bst = (uint32_t)probe_pad("XPNG_BURN_ST");
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_burn_tr), &btr, 4);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_burn_st), &bst, 4);
    }
#endif
    // (c->stream is created on first use, ctx_stream(): a caller that always passes its own stream never needs it, and every
    //  stream alive takes one of the runtime's hardware queues - with more streams than queues, streams share queues and serialise)
    if (hipHostMalloc((void **)&c->h_total, (uint64_t)batch * 8 + 64) != hipSuccess ||
        hipMemcpy(c->d_tiles, all.data(), VN * sizeof(TileDesc), hipMemcpyHostToDevice) != hipSuccess) {
        xpnghip_ctx_destroy(c);
        return fail("context setup failed");
    }
    *out = c;
    return 0;
}
extern "C" int xpnghip_ctx_create_batch(xpnghip_ctx **out, int device, uint64_t w, uint64_t h, int pxsz, uint32_t batch) {
    return xpnghip_ctx_create_range(out, device, w, h, pxsz, batch, 0, ~0ull);
}
extern "C" int xpnghip_ctx_create(xpnghip_ctx **out, int device, uint64_t w, uint64_t h, int pxsz) {
    return xpnghip_ctx_create_range(out, device, w, h, pxsz, 1, 0, ~0ull);
}

extern "C" uint64_t xpnghip_ctx_tile_count(const xpnghip_ctx *c) { return c ? c->tiles.size() : 0; }
extern "C" uint32_t xpnghip_ctx_batch(const xpnghip_ctx *c) { return c ? c->B : 0; }
extern "C" int xpnghip_ctx_tile(const xpnghip_ctx *c, uint64_t i, uint64_t xywh[4]) {
    if (!c || i >= c->tiles.size()) return 1;
    xywh[0] = c->tiles[i].x; xywh[1] = c->tiles[i].y; xywh[2] = c->tiles[i].w; xywh[3] = c->tiles[i].h;
    return 0;
}
extern "C" uint64_t xpnghip_ctx_blob_bound(const xpnghip_ctx *c, uint64_t t0, uint64_t t1) {
    uint64_t b = 0;
    for (uint64_t i = t0; c && i < t1 && i < c->tiles.size(); i++) b += (uint64_t)c->tiles[i].n * c->pxsz + 4;
    return b + 16;
}
extern "C" uint64_t xpnghip_ctx_workspace_bytes(const xpnghip_ctx *c) { return c ? c->ws_bytes : 0; }

static int check_range(const xpnghip_ctx *c, uint64_t t0, uint64_t t1) {
    if (!c) return fail("null context");
    if (t0 >= t1 || t1 > c->tiles.size()) return fail("bad tile range");
    if (t0 < c->r0 || t1 > c->r1) return fail("tile range outside the range this context was created for");
    return 0;
}

// upload the per-image pointer tables (only when they changed: the copy comes from pageable host memory)
static int set_ptrs(xpnghip_ctx *c, const void *const *in, void *const *outp, uint32_t nimg, hipStream_t s, bool dec = false) {
    if (nimg < 1 || nimg > c->B) return fail("batch size exceeds the context's batch");
    std::vector<const void *> &hin = dec ? c->h_dec_in_ptrs : c->h_in_ptrs;
    std::vector<void *> &hout = dec ? c->h_dec_out_ptrs : c->h_out_ptrs;
    bool same = hin.size() == nimg && hout.size() == nimg;
    for (uint32_t b = 0; same && b < nimg; b++) same = hin[b] == in[b] && hout[b] == outp[b];
    if (same) return 0;
    for (uint32_t b = 0; b < nimg; b++)
        if (((uintptr_t)(dec ? outp[b] : in[b]) & 15) || ((uintptr_t)(dec ? in[b] : outp[b]) & 3)) return fail("device buffers must be 16-byte aligned");
    HIPCHK(hipMemcpyAsync(dec ? c->d_dec_in_ptrs : c->d_in_ptrs, in, (uint64_t)nimg * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dec ? c->d_dec_out_ptrs : c->d_out_ptrs, outp, (uint64_t)nimg * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    hin.assign(in, in + nimg);
    hout.assign(outp, outp + nimg);
    return 0;
}

// the tile-major, size-sorted enumeration applies to launches over the context's whole tile range (XPNG_IMAGE_MAJOR=1: off)
static const uint32_t *order_for(const xpnghip_ctx *c, uint32_t t0, uint32_t t1) {
    return (t0 == c->r0 && t1 == c->r1 && !probe_env("XPNG_IMAGE_MAJOR")) ? c->d_order : nullptr;
}

// Level-1 stream scratch (k, block slots, context streams: 7.5 B/px, common.hpp) and the decode's symbol / residual planes
// (8 B/px: DecodeWs::arena) are ONE buffer: a context's encode intermediates are dead by the time the same context decodes.
static uint64_t scratch_bytes(const xpnghip_ctx *c) {
    const uint64_t enc = c->scratch_img * c->B + 8192, dec = 8 * c->plane_stride + (2u << 20);
    return enc > dec ? enc : dec;
}
static int ensure_scratch(xpnghip_ctx *c) {
    if (c->d_scratch) return 0;
    const uint64_t bytes = scratch_bytes(c);
    HIPCHK(hipMalloc((void **)&c->d_scratch, bytes));
    c->ws_bytes += bytes;
    return 0;
}
static int ensure_arena(xpnghip_ctx *c) {
    if (c->dec.arena) return 0;
    if (ensure_scratch(c)) return 1;
    c->dec.arena = c->d_scratch; c->dec.arena_bytes = scratch_bytes(c);
    return 0;
}
// the symbol planes (nl, r, g, b, + alpha symbols for RGBA): allocated on first use
static int ensure_planes(xpnghip_ctx *c) {
    if (c->d_planes) return 0;
    const uint64_t np = c->pxsz == 4 ? 5 : 4;  // nl, r, g, b (+ alpha symbols)
    HIPCHK(hipMalloc((void **)&c->d_planes, np * c->plane_stride + 8192));
    c->ws_bytes += np * c->plane_stride + 8192;
    return 0;
}

// predictor chooser (pp_rgbx).  Launch only.
// (probe builds: occupancy throttles of the bandwidth kernels in the pipelined paths, bytes of unused dynamic LDS per workgroup)
template <int PXSZ>
static int launch_chooser(xpnghip_ctx *c, uint32_t nimg, uint32_t t0, uint32_t t1, hipStream_t s, size_t pad = 0) {
    const uint32_t cnt = t1 - t0, total = nimg * cnt;
    const TileSel sel{t0, cnt, (uint32_t)c->tiles.size(), nimg, nullptr};
    const uint64_t bpr = c->W * PXSZ;
    XPNG_REQUIRE(c->d_in_ptrs, c->d_tiles, c->d_sums);
    if (dbg_skip("chooser")) return 0;
    if (t0 == 0 && t1 == c->tiles.size()) HIPCHK(hipMemsetAsync(c->d_sums, 0, (uint64_t)nimg * sel.N * 16, s));
    else for (uint32_t b = 0; b < nimg; b++) HIPCHK(hipMemsetAsync(c->d_sums + ((uint64_t)b * sel.N + t0) * 4, 0, (uint64_t)cnt * 16, s));
    const uint32_t strips = 16;
    k_chooser<PXSZ><<<total * strips, 256, pad, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, strips, c->d_sums);
    return 0;
}

// chooser + transform (BASELINE config 2).  Launch only; no sync.  d_in_ptrs already holds the raster pointers.
// hist: the transform also leaves the histogram of the nl plane in d_nlh (the stream lengths of the routing kernels come from it)
template <int PXSZ>
static int launch_transform(xpnghip_ctx *c, uint32_t nimg, uint32_t t0, uint32_t t1, hipStream_t s, size_t pad = 0, bool hist = false) {
    if (ensure_planes(c)) return 1;
    const uint32_t cnt = t1 - t0, total = nimg * cnt;
    // image-major here: these two kernels stream the rasters, and neighbouring workgroups on neighbouring rows of ONE raster
    // keep HBM pages open (measured with 64 distinct rasters: 2.25 ms per launch against 2.4-2.6 tile-major)
    const TileSel sel{t0, cnt, (uint32_t)c->tiles.size(), nimg, nullptr};
    const uint64_t bpr = c->W * PXSZ;
    uint32_t max_n = 0;
    for (uint32_t i = t0; i < t1; i++) max_n = c->tiles[i].n > max_n ? c->tiles[i].n : max_n;
    if (launch_chooser<PXSZ>(c, nimg, t0, t1, s, pad)) return 1;
    XPNG_REQUIRE(c->d_planes, c->d_nlh);
    uint32_t *nlh = hist ? c->d_nlh : nullptr;
    uint32_t max_w = 0, max_h = 0;
    for (uint32_t i = t0; i < t1; i++) { max_w = c->tiles[i].w > max_w ? c->tiles[i].w : max_w; max_h = c->tiles[i].h > max_h ? c->tiles[i].h : max_h; }
    if (PXSZ == 4 && max_w <= TR_MAXW && !probe_env("XPNG_GENERIC_TRANSFORM")) {
        const uint32_t spt_ = (max_h + TR_ROWS - 1) / TR_ROWS;
        if (!dbg_skip("transform")) k_m1_transform_rgba<<<(total * spt_ + 7) & ~7u, 256, pad, s>>>(c->d_in_ptrs, bpr, c->W * c->H * 4, c->d_tiles, sel, spt_, c->d_sums, c->d_planes, c->plane_stride, total * spt_, nlh, c->nlh_slots);
        c->nlh_generic = 0;
    } else if (PXSZ == 3 && max_w <= TR_MAXW && !probe_env("XPNG_GENERIC_TRANSFORM")) {
        const uint32_t spt_ = (max_h + TR_ROWS - 1) / TR_ROWS;
        k_m1_transform_rgb<<<(total * spt_ + 7) & ~7u, 256, pad, s>>>(c->d_in_ptrs, bpr, c->W * c->H * 3, c->d_tiles, sel, spt_, c->d_sums, c->d_planes, c->plane_stride, total * spt_, nlh, c->nlh_slots);
        c->nlh_generic = 0;
    } else {
        const uint32_t bpt = (max_n + 1024 * TG_REPS - 1) / (1024 * TG_REPS);
        k_m1_transform_generic<PXSZ><<<total * bpt, 256, 0, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, bpt, c->d_sums, c->d_planes, c->plane_stride, nlh, c->nlh_slots);
        c->nlh_generic = 1;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int xpnghip_m1_transform_device(xpnghip_ctx *c, const void *d_raster, uint64_t t0, uint64_t t1, void *stream) {
    if (check_range(c, t0, t1)) return 1;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx_stream(c);
    void *dummy = c->d_out_ptrs;  // no output buffer in this stage
    if (set_ptrs(c, &d_raster, &dummy, 1, s)) return 1;
    return c->pxsz == 4 ? launch_transform<4>(c, 1, (uint32_t)t0, (uint32_t)t1, s, 0, true) : launch_transform<3>(c, 1, (uint32_t)t0, (uint32_t)t1, s, 0, true);  // (with the nl histogram: the kernel the encode runs)
}

extern "C" int xpnghip_m1_transform_device_batch(xpnghip_ctx *c, const void *const *d_rasters, uint32_t nimg, uint64_t t0, uint64_t t1, void *stream) {
    if (check_range(c, t0, t1)) return 1;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx_stream(c);
    std::vector<void *> dummy(nimg, (void *)c->d_out_ptrs);  // no output buffers in this stage
    if (set_ptrs(c, d_rasters, dummy.data(), nimg, s)) return 1;
    return c->pxsz == 4 ? launch_transform<4>(c, nimg, (uint32_t)t0, (uint32_t)t1, s, 0, true) : launch_transform<3>(c, nimg, (uint32_t)t0, (uint32_t)t1, s, 0, true);
}

template <int PXSZ>
static int launch_encode_m1(xpnghip_ctx *c, uint32_t nimg, uint32_t t0, uint32_t t1, hipStream_t s) {
    dbg_count_sequence();
    const uint32_t cnt = t1 - t0, total = nimg * cnt;
    const TileSel sel{t0, cnt, (uint32_t)c->tiles.size(), nimg, order_for(c, t0, t1)};
    const uint64_t bpr = c->W * PXSZ;
    uint32_t max_w = 0;
    for (uint32_t i = t0; i < t1; i++) max_w = c->tiles[i].w > max_w ? c->tiles[i].w : max_w;
    const bool narrow = getenv("XPNG_NARROW_RANS") || (total * c->spt <= 2048 && !getenv("XPNG_WIDE_RANS"));
    const bool small_wg = (uint64_t)total * c->spt > 2048 && !probe_env("XPNG_BIG_BLOCKS");
    static const size_t pad_tr = probe_pad("XPNG_PAD_TR"), pad_st = probe_pad("XPNG_PAD_ST"), pad_ga = probe_pad("XPNG_PAD_GA");
    if (ensure_planes(c) || ensure_scratch(c)) return 1;  // (before their address is taken below)
    XPNG_REQUIRE(c->d_planes, c->d_in_ptrs, c->d_out_ptrs, c->d_tiles, c->d_scratch, c->d_sums, c->d_ctx_n, c->d_k_n,
                 c->d_blk_sz, c->d_tile_sz, c->d_tile_hdr, c->d_off, c->d_totals, c->d_wprep, c->d_wtab, c->d_wF, c->h_total);
    const uint8_t *planesA = c->d_planes;
    uint8_t *const watab = c->d_wtab + (((uint64_t)c->tiles.size() * c->B * WTAB_TILE_BYTES + 4096 + 511) & ~511ull);
    const bool alpha_side = !narrow && PXSZ == 4;
    if (alpha_side && !c->enc_side) HIPCHK(chain_stream_create(&c->enc_side));
    if (alpha_side && !c->ev_enc_fork) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_enc_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_enc_join, hipEventDisableTiming));
    }
    // (A fused transform + routing kernel - no nl / r / g / b planes, 3.7 B/px less HBM traffic - existed through round 3 behind
    //  XPNG_FUSED: one long-lived 28 KB workgroup per tile, 12 % slower in the pipeline every time it was measured, and it cannot
    //  know the stream lengths before it routes.  Removed with the worst-case stream layout; git history has it.)
    if (launch_transform<PXSZ>(c, nimg, t0, t1, s, pad_tr, true)) return 1;
    // Wide form, RGBA: the alpha chains are the longest serial stage of the encode and need only the alpha plane, so their
    // preparation and the chains themselves run on their own stream behind the transform
    // Size classes of the alpha chains: an experiment of round 4 that LOST (probe builds: XPNG_ENC_SPLIT=1).  The wide chain kernel lasts
    // as long as its longest chain (the biggest tile: 148 k steps of ~400 cycles at 4096^2; the median wavefront of a launch ends after two
    // thirds of that) and the wave-per-stream form steps in ~176 cycles, so the alpha streams of the biggest size class (the first n_big
    // tiles of the sorted enumeration x every image) were given a wavefront each on a third stream, k_rans2_encode with c_first = 9.
    // Bit-exact, 67 GPU tests green - and 33.7-35.3 against 38.2-40.1 Gpx/s at 64 x 4 (1088 such wavefronts of 26.8 KB of LDS and 176
    // registers each per launch: two thirds of the chip's LDS for ~11 ms), 42.0 against 44.3 at 128 x 4 (only the 128 tiles as large as
    // the largest).  profiles/r04_experiments.txt.
    uint32_t jb = 0;
    if (alpha_side && sel.order && probe_env("XPNG_ENC_SPLIT")) {
        const uint32_t nb = (uint64_t)c->n_big * nimg <= 1152 ? c->n_big : ((uint64_t)c->n_top * nimg <= 1152 ? c->n_top : 0u);
        if (nb > 0 && nb < cnt) jb = nb * nimg;
    }
    if (jb && !c->enc_side2) {
        HIPCHK(chain_stream_create(&c->enc_side2));
        HIPCHK(hipEventCreateWithFlags(&c->ev_enc_join2, hipEventDisableTiming));
    }
    if (alpha_side) {
        HIPCHK(hipEventRecord(c->ev_enc_fork, s));
        hipStream_t as = probe_env("XPNG_ONE_STREAM") ? s : c->enc_side;  // (probe builds: every kernel of the context on the caller's stream)
        HIPCHK(hipStreamWaitEvent(as, c->ev_enc_fork, 0));
        if (jb) {
            HIPCHK(hipStreamWaitEvent(c->enc_side2, c->ev_enc_fork, 0));
            if (!dbg_skip("chain_a")) k_rans2_encode<<<jb, 64, 0, c->enc_side2>>>(c->d_tiles, sel, 9, 1, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_blk_sz, c->d_wprep, nullptr);
            HIPCHK(hipEventRecord(c->ev_enc_join2, c->enc_side2));
        }
        if (!dbg_skip("prep_a")) k_rans2_prep<<<total - jb, 64, 0, as>>>(c->d_tiles, sel, 9, 1, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_blk_sz, c->d_wprep, c->d_wtab, c->d_wF, jb, watab);
        if (!dbg_skip("chain_a")) k_rans2_chain2<true><<<(total - jb + 31) / 32, 64, chain2_lds_bytes<true>() + probe_pad("XPNG_PAD_CHAIN"), as>>>(c->d_tiles, sel, total, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_wprep, c->d_wtab, jb, watab);
        HIPCHK(hipEventRecord(c->ev_enc_join, as));
    }
    // stream lengths (from the histogram of the nl plane the transform took as it wrote it) -> places of the nine context streams -> routing
    k_m1_lens<<<total, 64, 0, s>>>(c->d_nlh, c->nlh_slots, c->nlh_generic, c->d_tiles, sel, c->d_ctx_n);
    if (dbg_skip("streams")) {} else if (small_wg) k_m1_streams<PXSZ, 256><<<total, 256, pad_st, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, c->d_planes, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_k_n);
    else k_m1_streams<PXSZ, 1024><<<total, 1024, 0, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, c->d_planes, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_k_n);
    if (narrow) {
        // one wave per (tile, stream): fewer instructions per step (scalar cursors), best latency while every pair gets its own wave slot
        k_rans2_encode<<<total * c->spt, 64, 0, s>>>(c->d_tiles, sel, 0, c->spt, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_blk_sz, nullptr, c->stamps ? c->d_dbg : nullptr);
    } else {                           // every lane a chain: prep -> chain -> finish
        if (!dbg_skip("prep_c")) k_rans2_prep<<<total * 9, 64, 0, s>>>(c->d_tiles, sel, 0, 9, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_blk_sz, c->d_wprep, c->d_wtab, c->d_wF, 0, watab);
        if (!dbg_skip("chain_c")) k_rans2_chain2<false><<<((total + 31) / 32) * 9, 64, chain2_lds_bytes<false>() + probe_pad("XPNG_PAD_CHAIN"), s>>>(c->d_tiles, sel, total, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_wprep, c->d_wtab, 0, watab);
        if (alpha_side) HIPCHK(hipStreamWaitEvent(s, c->ev_enc_join, 0));
        if (jb) HIPCHK(hipStreamWaitEvent(s, c->ev_enc_join2, 0));
        if (!dbg_skip("finish")) k_rans2_finish<<<total * c->spt, 64, 0, s>>>(c->d_tiles, sel, c->spt, planesA, c->plane_stride, c->d_scratch, c->d_ctx_n, c->d_blk_sz, c->d_wprep, c->d_wF);
    }
    k_tile_sizes<<<(total + 255) / 256, 256, 0, s>>>(c->d_tiles, sel, total, PXSZ, c->spt, c->d_sums, c->d_k_n, c->d_blk_sz, c->d_tile_sz, c->d_tile_hdr);
    k_tile_offsets<<<nimg, 256, 0, s>>>(c->d_tile_sz, cnt, c->d_off, c->d_totals);
    if (!dbg_skip("gather")) k_tile_gather<<<total, 256, pad_ga, s>>>(c->d_in_ptrs, bpr, PXSZ, c->d_tiles, sel, c->spt, c->d_scratch, c->d_k_n, c->d_ctx_n, c->d_blk_sz, c->d_tile_hdr, c->d_off, c->d_out_ptrs);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_total, c->d_totals, (uint64_t)nimg * 8, hipMemcpyDeviceToHost, s));
    return 0;
}

static int ensure_m2(xpnghip_ctx *c) {
    if (c->d_scratch2) return 0;
    const uint64_t N = c->tiles.size(), VN = N * c->B;
    std::vector<uint64_t> sb(VN);
    uint64_t o = 0;
    for (uint64_t v = 0; v < VN; v++) { sb[v] = o; if (v % N >= c->r0 && v % N < c->r1) o += m2_tile_scratch(c->tiles[v % N].n); }
    if (hipMalloc((void **)&c->d_scratch2, o + 8192) != hipSuccess || hipMalloc((void **)&c->d_sbase2, VN * 8) != hipSuccess ||
        hipMalloc((void **)&c->d_flags2, VN * 4) != hipSuccess || hipMalloc((void **)&c->d_stream_n2, VN * M2_SLOTS * 4) != hipSuccess ||
        hipMalloc((void **)&c->d_blk2, VN * M2_SLOTS * sizeof(M2Blk)) != hipSuccess || hipMalloc((void **)&c->d_mt2, VN * sizeof(M2Tile)) != hipSuccess ||
        hipMalloc((void **)&c->d_info2, VN * sizeof(M2DecTile)) != hipSuccess || hipMalloc((void **)&c->d_tabs2, VN * M2_SLOTS * 512) != hipSuccess)
        return fail("hipMalloc failed (mode-2 workspace)");
    c->ws_bytes += o + 8192;
    HIPCHK(hipMemcpy(c->d_sbase2, sb.data(), VN * 8, hipMemcpyHostToDevice));
    return 0;
}

// mode 2: RGB only (libxpng.c:755 sends RGBA to mode 1 before the tile stage)
static int launch_encode_m2(xpnghip_ctx *c, uint32_t nimg, uint32_t t0, uint32_t t1, hipStream_t s) {
    if (ensure_m2(c)) return 1;
    const uint32_t cnt = t1 - t0, total = nimg * cnt;
    const TileSel sel{t0, cnt, (uint32_t)c->tiles.size(), nimg, nullptr};
    const uint64_t bpr = c->W * 3, VN = (uint64_t)c->B * sel.N;
    uint32_t max_n = 0;
    for (uint32_t i = t0; i < t1; i++) max_n = c->tiles[i].n > max_n ? c->tiles[i].n : max_n;
    XPNG_REQUIRE(c->d_scratch2, c->d_sbase2, c->d_flags2, c->d_stream_n2, c->d_blk2, c->d_mt2, c->d_in_ptrs, c->d_out_ptrs, c->d_tiles, c->d_sums, c->d_tile_sz, c->d_off,
                 c->d_totals, c->h_total);
    HIPCHK(hipMemsetAsync(c->d_flags2, 0, VN * 4, s));
    HIPCHK(hipMemsetAsync(c->d_stream_n2, 0, VN * M2_SLOTS * 4, s));
    HIPCHK(hipMemsetAsync(c->d_blk2, 0, VN * M2_SLOTS * sizeof(M2Blk), s));
    k_m2_classify<<<total * 16, 256, 0, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, 16, c->d_flags2);
    if (launch_transform<3>(c, nimg, t0, t1, s, 0, true)) return 1;  // chooser (PXSZ = 3, libxpng.c:663) + residual planes (allocates them on first use) + nl histogram
    XPNG_REQUIRE(c->d_planes);
    k_m2_count<<<total, 64, 0, s>>>(c->d_tiles, sel, c->d_flags2, c->d_nlh, c->nlh_slots, c->nlh_generic, c->d_stream_n2);  // stream lengths -> where every stream goes
    if ((uint64_t)total * M2_STREAMS > 2048 && !probe_env("XPNG_BIG_BLOCKS")) k_m2_streams<256><<<total, 256, 0, s>>>(c->d_tiles, sel, c->d_flags2, c->d_planes, c->plane_stride, c->d_scratch2, c->d_sbase2, c->d_stream_n2);
    else k_m2_streams<1024><<<total, 1024, 0, s>>>(c->d_tiles, sel, c->d_flags2, c->d_planes, c->plane_stride, c->d_scratch2, c->d_sbase2, c->d_stream_n2);
    const uint32_t gbpt = (max_n + 256 * M2_GRAY_REPS - 1) / (256 * M2_GRAY_REPS);
    k_m2_gray_syms<<<total * gbpt, 256, 0, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, gbpt, c->d_flags2, c->d_scratch2, c->d_sbase2, c->d_stream_n2);
    if (getenv("XPNG_NARROW_RANS") || ((uint64_t)total * M2_STREAMS <= 2048 && !getenv("XPNG_WIDE_RANS"))) {
        k_rans1_encode<<<total * M2_SLOTS, 64, 0, s>>>(c->d_tiles, sel, c->d_flags2, c->d_scratch2, c->d_sbase2, c->d_stream_n2, c->d_blk2);
    } else {  // every lane a chain: prep -> chains (small and big alphabets) -> finish
        if (!c->d_w1prep) {
            if (hipMalloc((void **)&c->d_w1prep, VN * M2_SLOTS * sizeof(W1Prep)) != hipSuccess ||
                hipMalloc((void **)&c->d_w1tab, VN * M2_SLOTS * W1_TAB_BYTES) != hipSuccess ||
                hipMalloc((void **)&c->d_w1F, VN * M2_SLOTS * 512) != hipSuccess)
                return fail("hipMalloc failed (mode-2 wide encode workspace)");
        }
        XPNG_REQUIRE(c->d_w1prep, c->d_w1tab, c->d_w1F);
        k_rans1_prep<<<total * M2_SLOTS, 64, 0, s>>>(c->d_tiles, sel, c->d_flags2, c->d_scratch2, c->d_sbase2, c->d_stream_n2, c->d_blk2, c->d_w1prep, c->d_w1tab, c->d_w1F);
        k_rans1_chain<true><<<((total + 15) / 16) * W1_BIG_SLOTS, 64, rans1_chain_lds_bytes<true>(), s>>>(c->d_tiles, sel, total, c->d_scratch2, c->d_sbase2, c->d_stream_n2, c->d_w1prep, c->d_w1tab);
        k_rans1_chain<false><<<((total + 31) / 32) * W1_SMALL_SLOTS, 64, rans1_chain_lds_bytes<false>(), s>>>(c->d_tiles, sel, total, c->d_scratch2, c->d_sbase2, c->d_stream_n2, c->d_w1prep, c->d_w1tab);
        k_rans1_finish<<<total * M2_SLOTS, 64, 0, s>>>(c->d_tiles, sel, c->d_scratch2, c->d_sbase2, c->d_stream_n2, c->d_blk2, c->d_w1prep, c->d_w1F);
    }
    k_m2_select<<<(total + 255) / 256, 256, 0, s>>>(c->d_tiles, sel, total, c->d_flags2, c->d_blk2, c->d_mt2, c->d_tile_sz);
    k_m2_bits<<<total, 256, 0, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, c->d_mt2, c->d_blk2, c->d_scratch2, c->d_sbase2, c->d_stream_n2);
    k_tile_offsets<<<nimg, 256, 0, s>>>(c->d_tile_sz, cnt, c->d_off, c->d_totals);
    k_m2_gather<<<total, 256, 0, s>>>(c->d_in_ptrs, bpr, c->d_tiles, sel, c->d_sums, c->d_mt2, c->d_blk2, c->d_scratch2, c->d_sbase2, c->d_off, c->d_out_ptrs);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_total, c->d_totals, (uint64_t)nimg * 8, hipMemcpyDeviceToHost, s));
    return 0;
}

extern "C" int xpnghip_encode_device_batch(xpnghip_ctx *c, int mode, const void *const *d_rasters, uint32_t nimg, uint64_t t0,
                                           uint64_t t1, void *const *d_blobs, uint64_t *blobs_len, void *stream) {
    if (check_range(c, t0, t1)) return 1;
    if (mode != 1 && mode != 2) return fail("tile mode must be 1 or 2");
    if (mode == 2 && c->pxsz != 3) return fail("mode 2 codes RGB only (the driver sends RGBA to mode 1, libxpng.c:755)");
    if (c->pxsz == 4)
        for (uint64_t i = t0; i < t1; i++)
            if (c->tiles[i].w < 4 || c->tiles[i].h < 4) return fail("RGBA tile narrower than 4 px: undefined in the reference; store level 7");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx_stream(c);
    if (set_ptrs(c, d_rasters, d_blobs, nimg, s)) return 1;
    const int rc = mode == 2 ? launch_encode_m2(c, nimg, (uint32_t)t0, (uint32_t)t1, s)
                   : c->pxsz == 4 ? launch_encode_m1<4>(c, nimg, (uint32_t)t0, (uint32_t)t1, s) : launch_encode_m1<3>(c, nimg, (uint32_t)t0, (uint32_t)t1, s);
    if (rc) return rc;
    if (blobs_len) {
        HIPCHK(hipStreamSynchronize(s));
        for (uint32_t b = 0; b < nimg; b++) blobs_len[b] = c->h_total[b];
    }
    return 0;
}
extern "C" int xpnghip_encode_device(xpnghip_ctx *c, int mode, const void *d_raster, uint64_t t0, uint64_t t1,
                                     void *d_blobs, uint64_t *blobs_len, void *stream) {
    return xpnghip_encode_device_batch(c, mode, &d_raster, 1, t0, t1, &d_blobs, blobs_len, stream);
}
extern "C" uint64_t xpnghip_ctx_last_blobs_len(xpnghip_ctx *c) { return c ? c->h_total[0] : 0; }
extern "C" uint64_t xpnghip_ctx_last_blobs_len_at(xpnghip_ctx *c, uint32_t img) { return c && img < c->B ? c->h_total[img] : 0; }

extern "C" int xpnghip_decode_device_batch(xpnghip_ctx *c, int mode, const void *const *d_blobs, const uint64_t *blobs_len, uint32_t nimg,
                                           const uint64_t *tile_off, uint64_t t0, uint64_t t1, void *const *d_rasters, void *stream) {
    if (check_range(c, t0, t1)) return 1;
    if (mode != 1 && mode != 2) return fail("tile mode must be 1 or 2");
    if (mode == 2 && c->pxsz != 3) return fail("mode 2 codes RGB only");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx_stream(c);
    if (set_ptrs(c, d_blobs, d_rasters, nimg, s, true)) return 1;
    if (!blobs_len) return fail("blob lengths are required (tile headers are validated against them)");
    if (c->h_blob_len.size() != nimg || memcmp(c->h_blob_len.data(), blobs_len, (size_t)nimg * 8) != 0) {
        HIPCHK(hipMemcpyAsync(c->d_blob_len, blobs_len, (uint64_t)nimg * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        c->h_blob_len.assign(blobs_len, blobs_len + nimg);
    }
    HIPCHK(hipMemsetAsync(c->d_status, 0, 4, s));
    uint32_t max_w = 0, max_h = 0, min_w = ~0u;
    for (uint64_t i = t0; i < t1; i++) { max_w = c->tiles[i].w > max_w ? c->tiles[i].w : max_w; max_h = c->tiles[i].h > max_h ? c->tiles[i].h : max_h; min_w = c->tiles[i].w < min_w ? c->tiles[i].w : min_w; }
    if (ensure_arena(c)) return 1;
    if (!c->dec.side) {  // (also after decode_ws_prepare rebuilt the workspace: it forgets the loan)
        if (!c->enc_side) HIPCHK(chain_stream_create(&c->enc_side));
        c->dec.side = c->enc_side; c->dec.side_borrowed = true;
    }
    if (probe_env("XPNG_ONE_STREAM")) { c->dec.side = s; c->dec.side_borrowed = true; }
    XPNG_REQUIRE(c->d_dec_in_ptrs, c->d_dec_out_ptrs, c->d_blob_len, c->d_status, c->d_tiles, c->dec.arena);
    if (mode == 2) {
        if (ensure_m2(c)) return 1;
        XPNG_REQUIRE(c->d_info2, c->d_blk2, c->d_tabs2, c->d_scratch2, c->d_sbase2, c->d_stream_n2);
        return decode_m2_launch(c->dec, nimg, c->tiles.size(), c->plane_stride, c->d_tiles, c->W, max_w, max_h, c->d_dec_in_ptrs, c->d_blob_len, c->d_status, tile_off, (uint32_t)t0,
                                (uint32_t)t1, c->d_dec_out_ptrs, c->d_info2, c->d_blk2, c->d_tabs2, c->d_scratch2, c->d_sbase2, c->d_stream_n2, s, g_err);
    }
    return decode_m1_launch(c->dec, nimg, c->tiles.size(), c->plane_stride, c->d_tiles, c->W, max_w, max_h, c->pxsz, c->d_dec_in_ptrs, c->d_blob_len, c->d_status, tile_off,
                            (uint32_t)t0, (uint32_t)t1, c->d_dec_out_ptrs, s, g_err, c->stamps ? c->d_dbg + c->tiles.size() * c->B * 80 : nullptr,
                            order_for(c, (uint32_t)t0, (uint32_t)t1), order_for(c, (uint32_t)t0, (uint32_t)t1) ? c->n_big : 0u, min_w);
}
extern "C" int xpnghip_decode_device(xpnghip_ctx *c, int mode, const void *d_blobs, uint64_t blobs_len,
                                     const uint64_t *tile_off, uint64_t t0, uint64_t t1, void *d_raster, void *stream) {
    return xpnghip_decode_device_batch(c, mode, &d_blobs, &blobs_len, 1, tile_off, t0, t1, &d_raster, stream);
}
// Synchronises `stream` and reports the last decode: 0 = every tile header was consistent, 1 = at least one tile was
// rejected (its pixels were left untouched), -1 = HIP error.
extern "C" int xpnghip_ctx_decode_status(xpnghip_ctx *c, void *stream) {
    if (!c) return -1;
    hipStream_t s = stream ? (hipStream_t)stream : ctx_stream(c);
    // (on the call's own stream, into pinned memory: a plain hipMemcpy runs on the null stream, which waits for every blocking
    //  stream of the device - a caller with several contexts in flight would wait for all of them here, as the pipeline shards of
    //  xpnghip_decode_tiles did: the two early shards "finished" when the last one did)
    volatile uint32_t *hv = reinterpret_cast<volatile uint32_t *>(c->h_total + c->B);  // (h_total has 64 spare bytes behind its B entries)
    if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync((void *)hv, c->d_status, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) return -1;
    return (int)(*hv & 1);
}

#include "wrappers.hpp"

// ---- introspection for parity tests ----------------------------------------------------------------------
// wave probe (common.hpp): register a device buffer of `cap` WaveProbe records (nullptr: off); the count so far
#ifdef XPNG_PROBES
extern "C" int xpnghip_debug_probe(void *d_buf, uint32_t cap) {
    WaveProbe *p = (WaveProbe *)d_buf;
    const uint32_t zero = 0;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_buf), &p, sizeof(p)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_cap), &cap, sizeof(cap)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_n), &zero, sizeof(zero)));
    return 0;
}
extern "C" int64_t xpnghip_debug_probe_count(void) {
    uint32_t n = 0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_probe_n), sizeof(n)) != hipSuccess) return -1;
    return n;
}
extern "C" int xpnghip_probes_built(void) { return 1; }
#else
extern "C" int xpnghip_debug_probe(void *, uint32_t) { return fail("the wave probe exists only in libxpng_hip_probes.so (make probes)"); }
extern "C" int64_t xpnghip_debug_probe_count(void) { return -1; }
extern "C" int xpnghip_probes_built(void) { return 0; }
#endif

extern "C" int64_t xpnghip_debug_fetch(xpnghip_ctx *c, int what, uint64_t tile, void *out, uint64_t cap) {
    if (!c || tile >= c->tiles.size() || !out) return -1;
    if (hipSetDevice(c->device) != hipSuccess || (c->stream && hipStreamSynchronize(c->stream) != hipSuccess)) return -1;  // (debug_fetch follows a call that synchronised its stream or used this one)
    const TileDesc &t = c->tiles[tile];
    const uint8_t *src = nullptr;
    uint64_t bytes = 0;
    uint32_t tmp[16];
    auto d2h = [&](void *dst, const void *s, uint64_t b) { return hipMemcpy(dst, s, b, hipMemcpyDeviceToHost) == hipSuccess; };
    if (what == 0) {
        if (!d2h(tmp, c->d_sums + tile * 4, 16)) return -1;
        int m = 0; uint32_t r = tmp[0];
        for (int k = 1; k < 4; k++) if (tmp[k] < r) { m = k; r = tmp[k]; }
        uint8_t pr = (t.w < 4 || t.h < 4) ? 0 : (uint8_t)((c->pxsz & 4) | m);
        if (cap < 1) return -1;
        *(uint8_t *)out = pr;
        return 1;
    } else if (what >= 1 && what <= 5) {
        if (!c->d_planes || (what == 5 && c->pxsz != 4)) return -1;  // the five planes exist after xpnghip_m1_transform_device (config-2 entry) or a mode-2 encode
        src = c->d_planes + (uint64_t)(what - 1) * c->plane_stride + t.pbase; bytes = t.n;
    } else if (what >= 10 && what <= 29 && !c->d_scratch) {
        return -1;  // no level-1 encode has run on this context
    } else if (what >= 10 && what <= 18) {
        if (!d2h(tmp, c->d_ctx_n + tile * 9, 36)) return -1;
        src = c->d_scratch + t.sbase + off_ctx(t.n, tmp, what - 10); bytes = tmp[what - 10];
    } else if (what == 19) {
        if (!d2h(tmp, c->d_k_n + tile, 4)) return -1;
        src = c->d_scratch + t.sbase + off_kw(t.n); bytes = 4ull * tmp[0];
    } else if (what >= 20 && what <= 29) {
        if ((uint32_t)(what - 20) >= c->spt) return 0;
        uint32_t cn[9];
        if (!d2h(cn, c->d_ctx_n + tile * 9, 36) || !d2h(tmp, c->d_blk_sz + tile * 10, 40)) return -1;
        src = c->d_scratch + t.sbase + off_blk(t.n, cn, what - 20); bytes = tmp[what - 20];
    } else if (what == 40 || what == 41) {
        if (!c->d_dbg) return -1;  // phase stamps exist in probe builds only
        src = (const uint8_t *)(c->d_dbg + ((what - 40) * c->tiles.size() * c->B + tile) * 80); bytes = 640;
    } else if (what == 30) {
        src = (const uint8_t *)(c->d_sums + tile * 4); bytes = 16;
    } else return -1;
    if (bytes > cap) return -1;
    if (bytes && !d2h(out, src, bytes)) return -1;
    return (int64_t)bytes;
}
