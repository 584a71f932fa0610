// wrappers.hpp -- the host-buffer entry points of include/xpng_hip.h (included by xpng_hip.hip): what the host C driver
// calls in place of the reference's two thread fan-outs (libxpng.c:758 + 764-769, 982-983).
//
//   * One process, up to T devices.  The reference's `T` is its worker count (libxpng.c:146-151: T = min(T, N) threads over a
//     shared tile cursor); here T devices each take one contiguous, pixel-weighted tile range (the same split as
//     xpng_amd/shard.py), encode / decode it from their own band of the raster, and the blob ranges are gathered on device 0 by
//     peer copies over xGMI for the one concatenation (libxpng.c:764-769) before the single copy to the host.
//   * Contexts (tile table + workspace per geometry, device and tile range) are kept in a small LRU, and every staging
//     buffer lives in its context and only ever grows: repeat calls allocate nothing.
//   * Everything runs on explicit devices (XPNG_DEVICE = first device, default 0) and the caller's current device is restored.
#pragma once

static std::mutex g_mu;

static int base_device() {
    static const int d = [] { const char *e = getenv("XPNG_DEVICE"); return e ? atoi(e) : 0; }();
    return d;
}
// XPNG_FAKE_DEVICES=n: rehearse the multi-device path on a box with one GPU (n shards, all on the base device)
static int fake_devices() {
    const char *e = getenv("XPNG_FAKE_DEVICES");
    return e ? atoi(e) : 0;
}
static int usable_devices() {
    const int f = fake_devices();
    if (f > 0) return xpnghip_device_count() > base_device() ? f : 0;
    const int n = xpnghip_device_count() - base_device();
    return n > 0 ? n : 0;
}
static int shard_device(int k) { return fake_devices() > 0 ? base_device() : base_device() + k; }

struct DevGuard {  // pins the work to our devices and hands the caller's current device back
    int prev = -1;
    DevGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static uint64_t tile_count_for(uint64_t W, uint64_t H) {
    if (W * H <= TILE_AREA) return 1;
    uint64_t bw, bh, nx, ny, a, b;
    if (W < 444) { bw = W; bh = TILE_AREA / W; }
    else if (H < 444) { bh = H; bw = TILE_AREA / H; }
    else bw = bh = 444;
    split_axis(W, bw, nx, a, b);
    split_axis(H, bh, ny, a, b);
    return nx * ny;
}

// ---- context cache -----------------------------------------------------------------------------------------
static std::vector<xpnghip_ctx *> g_lru;  // most recently used first
static uint64_t g_call = 0;               // contexts touched by the call in progress (ctx->call == g_call) are never evicted
constexpr size_t LRU_MAX = 12;
constexpr uint64_t LRU_MAX_BYTES = 64ull << 30;
static void lru_trim(size_t keep, uint64_t keep_bytes) {
    uint64_t sum = 0;
    std::vector<xpnghip_ctx *> kept;
    for (xpnghip_ctx *c : g_lru) {
        sum += c->ws_bytes;
        if (c->call == g_call || (kept.size() < keep && sum <= keep_bytes)) kept.push_back(c);
        else xpnghip_ctx_destroy(c);
    }
    g_lru.swap(kept);
}
static xpnghip_ctx *cached_ctx(int dev, uint64_t w, uint64_t h, int pxsz, uint64_t r0 = 0, uint64_t r1 = ~0ull) {
    for (size_t i = 0; i < g_lru.size(); i++) {
        xpnghip_ctx *c = g_lru[i];
        if (c->device == dev && c->W == w && c->H == h && c->pxsz == pxsz && c->r0 == r0 && c->r1 == std::min<uint64_t>(r1, c->tiles.size())) {
            g_lru.erase(g_lru.begin() + (long)i);
            g_lru.insert(g_lru.begin(), c);
            c->call = g_call;
            return c;
        }
    }
    xpnghip_ctx *c = nullptr;
    if (xpnghip_ctx_create_range(&c, dev, w, h, pxsz, 1, r0, r1)) {
        lru_trim(0, 0);  // out of memory, perhaps: give the cached workspaces back and try once more
        if (xpnghip_ctx_create_range(&c, dev, w, h, pxsz, 1, r0, r1)) return nullptr;
    }
    c->call = g_call;
    g_lru.insert(g_lru.begin(), c);
    lru_trim(LRU_MAX, LRU_MAX_BYTES);
    return c;
}
static int ensure_buf(uint8_t *&p, uint64_t &cap, uint64_t need) {
    if (cap >= need && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    HIPCHK(hipMalloc((void **)&p, need + 64));
    cap = need;
    return 0;
}

// ---- tile ranges of a multi-device call -----------------------------------------------------------------------
struct Shard { int dev; uint64_t r0, r1; uint32_t y0, y1; xpnghip_ctx *c; uint64_t len, off; };
// how many devices a call uses: T >= 1 -> min(T, usable, N); T == 0 -> as many as leave each device at least 256 tiles (a
// single image is bound by its longest entropy chain, not by tile count: spreading 81 tiles over 8 devices buys nothing)
static int devices_for(uint64_t T, uint64_t N) {
    const int have = usable_devices();
    if (have < 1) return 0;
    uint64_t d = T ? T : std::max<uint64_t>(1, N / 256);
    if (const char *e = getenv("XPNG_GPUS")) if (!T && atoi(e) > 0) d = (uint64_t)atoi(e);
    d = std::min<uint64_t>(d, (uint64_t)have);
    d = std::min<uint64_t>(d, N);
    return (int)std::max<uint64_t>(d, 1);
}
// contiguous ranges balanced by pixel count (xpng_amd/shard.py weighted_tile_ranges; the reference's cursor hands out tiles
// one by one, libxpng.c:150-151 - any partition gives the same bytes)
static std::vector<Shard> make_shards(const std::vector<TileDesc> &tiles, int D) {
    std::vector<Shard> out;
    uint64_t total = 0, acc = 0, start = 0;
    for (const TileDesc &t : tiles) total += t.n;
    int k = 1;
    for (uint64_t i = 0; i < tiles.size(); i++) {
        acc += tiles[i].n;
        while (k < D && acc * (uint64_t)D >= total * (uint64_t)k && i + 1 <= tiles.size() - (uint64_t)(D - k)) {
            out.push_back(Shard{0, start, i + 1, 0, 0, nullptr, 0, 0});
            start = i + 1; k++;
        }
    }
    out.push_back(Shard{0, start, tiles.size(), 0, 0, nullptr, 0, 0});
    for (size_t s = 0; s < out.size(); s++) {
        Shard &sh = out[s];
        sh.dev = shard_device((int)s);
        uint32_t y0 = ~0u, y1 = 0;
        for (uint64_t i = sh.r0; i < sh.r1; i++) { y0 = std::min(y0, tiles[i].y); y1 = std::max(y1, tiles[i].y + tiles[i].h); }
        sh.y0 = y0; sh.y1 = y1;
    }
    return out;
}

static int check_geometry(uint64_t w, uint64_t h, int pxsz) {
    if (!w || !h || w > (1u << 24) || h > (1u << 24) || (pxsz != 3 && pxsz != 4)) return fail("bad raster geometry");
    if (usable_devices() < 1) return fail("no usable HIP device (libxpng_hip has no CPU fallback)");
    return 0;
}

// Encode on D devices.  The raster is either in host memory (h_src) or already staged on the base device (d_src).
static int encode_multi(int D, int mode, const uint8_t *h_src, const uint8_t *d_src, uint64_t w, uint64_t h, int pxsz, uint8_t **blobs, uint64_t *blobs_len) {
    const uint64_t bpr = w * (uint64_t)pxsz;
    std::vector<TileDesc> tiles;
    build_tiles(w, h, tiles);
    std::vector<Shard> sh = make_shards(tiles, D);
    const int dev0 = sh[0].dev;
    // shard 0 encodes straight into the gather buffer (sized for the whole image), the others into their own
    uint64_t whole_bound = 16;
    for (const TileDesc &t : tiles) whole_bound += (uint64_t)t.n * pxsz + 4;
    for (size_t k = 0; k < sh.size(); k++) {
        Shard &s = sh[k];
        HIPCHK(hipSetDevice(s.dev));
        if (!(s.c = cached_ctx(s.dev, w, h, pxsz, s.r0, s.r1))) return 1;
        xpnghip_ctx *c = s.c;
        const uint64_t band = (uint64_t)(s.y1 - s.y0) * bpr;
        if (ensure_buf(c->d_raster, c->cap_raster, band + 16)) return 1;
        if (ensure_buf(c->d_blobs, c->cap_blobs, k == 0 ? whole_bound : xpnghip_ctx_blob_bound(c, s.r0, s.r1))) return 1;
        // kernels address rows absolutely: the band is handed over as if the whole raster were there (only rows [y0, y1) are
        // touched); it starts at the 16-byte phase row y0 has in the whole raster, so the virtual base stays 16-byte aligned
        uint8_t *bandp = c->d_raster + (((uint64_t)s.y0 * bpr) & 15);
        if (h_src) HIPCHK(hipMemcpyAsync(bandp, h_src + (uint64_t)s.y0 * bpr, band, hipMemcpyHostToDevice, c->stream));
        else if (s.dev == dev0) HIPCHK(hipMemcpyAsync(bandp, d_src + (uint64_t)s.y0 * bpr, band, hipMemcpyDeviceToDevice, c->stream));
        else HIPCHK(hipMemcpyPeerAsync(bandp, s.dev, d_src + (uint64_t)s.y0 * bpr, dev0, band, c->stream));
        if (xpnghip_encode_device(c, mode, bandp - (uint64_t)s.y0 * bpr, s.r0, s.r1, c->d_blobs, nullptr, nullptr)) return 1;
    }
    uint64_t total = 0;
    for (Shard &s : sh) {
        HIPCHK(hipSetDevice(s.dev));
        HIPCHK(hipStreamSynchronize(s.c->stream));
        s.len = s.c->h_total[0]; s.off = total; total += s.len;
    }
    // the one exchange of the path (libxpng.c:764-769): blob ranges -> device 0, behind shard 0's own bytes
    for (size_t k = 1; k < sh.size(); k++) {
        Shard &s = sh[k];
        HIPCHK(hipSetDevice(s.dev));
        if (s.dev == dev0) HIPCHK(hipMemcpyAsync(sh[0].c->d_blobs + s.off, s.c->d_blobs, s.len, hipMemcpyDeviceToDevice, s.c->stream));
        else HIPCHK(hipMemcpyPeerAsync(sh[0].c->d_blobs + s.off, dev0, s.c->d_blobs, s.dev, s.len, s.c->stream));
    }
    for (size_t k = 1; k < sh.size(); k++) { HIPCHK(hipSetDevice(sh[k].dev)); HIPCHK(hipStreamSynchronize(sh[k].c->stream)); }
    uint8_t *out = (uint8_t *)malloc(total ? total : 1);
    if (!out) return fail("malloc failed");
    HIPCHK(hipSetDevice(dev0));
    if (hipMemcpy(out, sh[0].c->d_blobs, total, hipMemcpyDeviceToHost) != hipSuccess) { free(out); return fail("blob download failed"); }
    *blobs = out; *blobs_len = total;
    return 0;
}

static int encode_tiles_impl(uint64_t T, int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz, uint8_t **blobs, uint64_t *blobs_len) {
    if (!raster || !blobs || !blobs_len) return fail("null argument");
    if (check_geometry(w, h, pxsz)) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    DevGuard guard;
    g_call++;
    const uint64_t N = tile_count_for(w, h);
    const int D = devices_for(T, N);
    if (D > 1) return encode_multi(D, mode, raster, nullptr, w, h, pxsz, blobs, blobs_len);
    HIPCHK(hipSetDevice(base_device()));
    xpnghip_ctx *c = cached_ctx(base_device(), w, h, pxsz);
    if (!c) return 1;
    const uint64_t s = w * h * (uint64_t)pxsz;
    if (ensure_buf(c->d_raster, c->cap_raster, s) || ensure_buf(c->d_blobs, c->cap_blobs, xpnghip_ctx_blob_bound(c, 0, N))) return 1;
    HIPCHK(hipMemcpyAsync(c->d_raster, raster, s, hipMemcpyHostToDevice, c->stream));
    uint64_t len = 0;
    if (xpnghip_encode_device(c, mode, c->d_raster, 0, N, c->d_blobs, &len, nullptr)) return 1;
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    if (!out) return fail("malloc failed");
    if (hipMemcpy(out, c->d_blobs, len, hipMemcpyDeviceToHost) != hipSuccess) { free(out); return fail("blob download failed"); }
    *blobs = out; *blobs_len = len;
    return 0;
}

// Decode on D devices: the host walks the tile sizes (the file is in host memory, libxpng.c:982), every device gets the blob
// range of its tiles and fills its band; the bands come back as one rectangle per tile row of the range.
static int decode_multi(int D, int mode, const uint8_t *blobs, const std::vector<uint64_t> &off, uint64_t w, uint64_t h, int pxsz, uint8_t *raster) {
    const uint64_t bpr = w * (uint64_t)pxsz;
    std::vector<TileDesc> tiles;
    build_tiles(w, h, tiles);
    std::vector<Shard> sh = make_shards(tiles, D);
    std::vector<uint64_t> rel;
    for (Shard &s : sh) {
        HIPCHK(hipSetDevice(s.dev));
        if (!(s.c = cached_ctx(s.dev, w, h, pxsz, s.r0, s.r1))) return 1;
        xpnghip_ctx *c = s.c;
        s.off = off[s.r0]; s.len = off[s.r1] - off[s.r0];
        const uint64_t band = (uint64_t)(s.y1 - s.y0) * bpr;
        if (ensure_buf(c->d_raster, c->cap_raster, band + 16) || ensure_buf(c->d_blob_in, c->cap_blob_in, s.len)) return 1;
        HIPCHK(hipMemcpyAsync(c->d_blob_in, blobs + s.off, s.len, hipMemcpyHostToDevice, c->stream));
        rel.assign(off.begin() + (long)s.r0, off.begin() + (long)s.r1);
        for (uint64_t &o : rel) o -= s.off;
        uint8_t *bandp = c->d_raster + (((uint64_t)s.y0 * bpr) & 15);
        if (xpnghip_decode_device(c, mode, c->d_blob_in, s.len, rel.data(), s.r0, s.r1, bandp - (uint64_t)s.y0 * bpr, nullptr)) return 1;
    }
    int rc = 0;
    for (Shard &s : sh) {
        HIPCHK(hipSetDevice(s.dev));
        const int st = xpnghip_ctx_decode_status(s.c, nullptr);
        if (st == 1) rc = fail("corrupt file: a tile header is inconsistent with the tile table");
        else if (st != 0) rc = fail("decode failed");
        if (rc) continue;
        for (uint64_t i = s.r0; i < s.r1;) {  // tiles i..j-1 share a tile row: one rectangle
            uint64_t j = i + 1;
            while (j < s.r1 && tiles[j].y == tiles[i].y) j++;
            const uint64_t x0 = tiles[i].x, x1 = tiles[j - 1].x + tiles[j - 1].w, y = tiles[i].y;
            if (hipMemcpy2DAsync(raster + y * bpr + x0 * pxsz, bpr, s.c->d_raster + (((uint64_t)s.y0 * bpr) & 15) + (y - s.y0) * bpr + x0 * pxsz, bpr, (x1 - x0) * pxsz, tiles[i].h,
                                 hipMemcpyDeviceToHost, s.c->stream) != hipSuccess) { rc = fail("raster download failed"); break; }
            i = j;
        }
    }
    for (Shard &s : sh) { (void)hipSetDevice(s.dev); if (hipStreamSynchronize(s.c->stream) != hipSuccess && !rc) rc = fail("raster download failed"); }
    return rc;
}

static int decode_tiles_impl(uint64_t T, int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h, int pxsz, uint8_t *raster) {
    if (!raster || !blobs) return fail("null argument");
    if (check_geometry(w, h, pxsz)) return 1;
    // before any workspace is sized from the header's claim: every tile needs at least its 4-byte size word
    const uint64_t N = tile_count_for(w, h);
    if (blobs_len / 4 < N) return fail("truncated file: shorter than its tile table");
    std::vector<uint64_t> off(N + 1);
    uint64_t o = 0;
    for (uint64_t i = 0; i < N; i++) {  // serial size walk, libxpng.c:982
        if (o + 4 > blobs_len) return fail("truncated file: tile table runs past the end");
        uint32_t h0; memcpy(&h0, blobs + o, 4);
        off[i] = o; o += h0 & 0xFFFFFF;
    }
    if (o > blobs_len) return fail("truncated file: last tile runs past the end");
    off[N] = o;
    std::lock_guard<std::mutex> lk(g_mu);
    DevGuard guard;
    g_call++;
    const int D = devices_for(T, N);
    if (D > 1) return decode_multi(D, mode, blobs, off, w, h, pxsz, raster);
    HIPCHK(hipSetDevice(base_device()));
    xpnghip_ctx *c = cached_ctx(base_device(), w, h, pxsz);
    if (!c) return 1;
    const uint64_t s = w * h * (uint64_t)pxsz;
    if (ensure_buf(c->d_raster, c->cap_raster, s) || ensure_buf(c->d_blob_in, c->cap_blob_in, blobs_len)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_blob_in, blobs, blobs_len, hipMemcpyHostToDevice, c->stream));
    if (xpnghip_decode_device(c, mode, c->d_blob_in, blobs_len, off.data(), 0, N, c->d_raster, nullptr)) return 1;
    // While the kernels run: touch the caller's (typically freshly malloc()ed) raster, one write per page, on a few threads.  Its
    // first-touch page faults (16 k of them for a 4096^2 RGBA raster: ~4 ms) otherwise land inside the download.
    if (s >= (32u << 20) && !getenv("XPNG_NO_PREFAULT")) {
        constexpr int NT = 4;
        std::thread th[NT];
        const uint64_t part = ((s / NT) + 4095) & ~4095ull;
        for (int t = 0; t < NT; t++) {
            const uint64_t a = std::min<uint64_t>(s, t * part), b = std::min<uint64_t>(s, (t + 1) * part);
            th[t] = std::thread([=] { for (uint64_t o = a; o < b; o += 4096) raster[o] = 0; });
        }
        for (int t = 0; t < NT; t++) th[t].join();
    }
    const int st = xpnghip_ctx_decode_status(c, nullptr);
    if (st == 1) return fail("corrupt file: a tile header is inconsistent with the tile table");
    if (st != 0) return fail("decode failed");
    // (a pinned staging buffer with chunked copies and host copy threads measured no better than this plain copy: 21.8 against 21.5 ms)
    HIPCHK(hipMemcpy(raster, c->d_raster, s, hipMemcpyDeviceToHost));
    return 0;
}

// extern "C" must not leak C++ exceptions (std::bad_alloc from a header that claims an absurd geometry)
#define XPNG_GUARDED(expr)                                               \
    try { return (expr); }                                               \
    catch (const std::bad_alloc &) { return fail("out of host memory"); } \
    catch (...) { return fail("unexpected C++ exception"); }

extern "C" int xpnghip_encode_tiles_T(uint64_t T, int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz, uint8_t **blobs, uint64_t *blobs_len) {
    XPNG_GUARDED(encode_tiles_impl(T, mode, raster, w, h, pxsz, blobs, blobs_len))
}
extern "C" int xpnghip_encode_tiles(int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz, uint8_t **blobs, uint64_t *blobs_len) {
    XPNG_GUARDED(encode_tiles_impl(1, mode, raster, w, h, pxsz, blobs, blobs_len))
}
extern "C" int xpnghip_decode_tiles_T(uint64_t T, int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h, int pxsz, uint8_t *raster) {
    XPNG_GUARDED(decode_tiles_impl(T, mode, blobs, blobs_len, w, h, pxsz, raster))
}
extern "C" int xpnghip_decode_tiles(int mode, const uint8_t *blobs, uint64_t blobs_len, uint64_t w, uint64_t h, int pxsz, uint8_t *raster) {
    XPNG_GUARDED(decode_tiles_impl(1, mode, blobs, blobs_len, w, h, pxsz, raster))
}

// ---- staged image (normalize_RGBA and the single-colour test on the device) ---------------------------
static uint32_t *g_flags = nullptr;  // 4 device words for the OR-reductions (base device)
static int norm_device(const void *d_rgba, uint64_t npx, void *d_out, int *pxsz_out, int *rewritten, hipStream_t s, uint32_t *flags) {
    HIPCHK(hipMemsetAsync(flags, 0, 16, s));
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((npx / 4 + 255) / 256 + 1, 256 * 16);
    k_norm_flags<<<blocks, 256, 0, s>>>((const uint32_t *)d_rgba, npx, flags);
    uint32_t f[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(f, flags, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *pxsz_out = 4; *rewritten = 0;
    if (f[0]) { k_norm_zero_hidden<<<blocks, 256, 0, s>>>((const uint32_t *)d_rgba, (uint32_t *)d_out, npx); *rewritten = 1; }
    else if (!f[1]) { k_norm_to_rgb<<<blocks, 256, 0, s>>>((const uint32_t *)d_rgba, (uint8_t *)d_out, npx); *pxsz_out = 3; *rewritten = 1; }
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int xpnghip_normalize_device(const void *d_rgba, uint64_t npx, void *d_out, int *pxsz_out, int *rewritten, void *stream) {
    if (!d_rgba || !d_out || !pxsz_out || !rewritten || !npx) return fail("null argument");
    if (((uintptr_t)d_rgba & 15) || ((uintptr_t)d_out & 3)) return fail("device buffers must be 16-byte aligned");
    // (the caller's raster lives on the caller's current device: the flag words are allocated there, per call)
    uint32_t *flags = nullptr;
    HIPCHK(hipMalloc((void **)&flags, 16));
    const int rc = norm_device(d_rgba, npx, d_out, pxsz_out, rewritten, (hipStream_t)stream, flags);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(flags);
    return rc;
}

static struct Staged {
    bool open = false;
    int prev_dev = -1;
    uint64_t w = 0, h = 0, cap_in = 0, cap_norm = 0;
    int pxsz = 0;
    uint8_t *d_in = nullptr, *d_norm = nullptr;  // uploaded raster; rewritten raster (when normalisation changed it)
    const uint8_t *cur = nullptr;                // the staged (normalised) raster
} g_img;

static int image_begin_impl(const uint8_t *raster, uint64_t w, uint64_t h, int pxsz_in, int *pxsz_out) {
    // (g_mu is held; on failure the caller releases it)
    if (hipGetDevice(&g_img.prev_dev) != hipSuccess) g_img.prev_dev = -1;
    HIPCHK(hipSetDevice(base_device()));
    const uint64_t s = w * h * (uint64_t)pxsz_in;
    if (ensure_buf(g_img.d_in, g_img.cap_in, s)) return 1;
    HIPCHK(hipMemcpy(g_img.d_in, raster, s, hipMemcpyHostToDevice));
    g_img.w = w; g_img.h = h; g_img.pxsz = pxsz_in; g_img.cur = g_img.d_in;
    if (pxsz_in == 4) {
        if (ensure_buf(g_img.d_norm, g_img.cap_norm, s)) return 1;
        if (!g_flags) HIPCHK(hipMalloc((void **)&g_flags, 16));
        int rewritten = 0;
        if (norm_device(g_img.d_in, w * h, g_img.d_norm, &g_img.pxsz, &rewritten, nullptr, g_flags)) return 1;
        if (rewritten) g_img.cur = g_img.d_norm;
    }
    *pxsz_out = g_img.pxsz;
    return 0;
}
extern "C" int xpnghip_image_begin(const uint8_t *raster, uint64_t w, uint64_t h, int pxsz_in, int *pxsz_out) {
    if (!raster || !pxsz_out) return fail("bad argument");
    if (check_geometry(w, h, pxsz_in)) return 1;
    g_mu.lock();
    const int rc = image_begin_impl(raster, w, h, pxsz_in, pxsz_out);
    if (rc) {
        if (g_img.prev_dev >= 0) (void)hipSetDevice(g_img.prev_dev);
        g_mu.unlock();
        return rc;
    }
    g_img.open = true;
    return 0;  // (the lock stays held until xpnghip_image_end)
}
extern "C" void xpnghip_image_end(void) {
    if (!g_img.open) return;
    g_img.open = false;
    if (g_img.prev_dev >= 0) (void)hipSetDevice(g_img.prev_dev);
    g_mu.unlock();
}
extern "C" int xpnghip_image_single_colour(int *single) {
    if (!g_img.open || !single) return fail("no staged image");
    HIPCHK(hipSetDevice(base_device()));
    const uint64_t n = g_img.w * g_img.h;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 16);
    if (!g_flags) HIPCHK(hipMalloc((void **)&g_flags, 16));
    HIPCHK(hipMemsetAsync(g_flags + 2, 0, 4, nullptr));
    if (g_img.pxsz == 4) k_any_differs<4><<<blocks, 256>>>(g_img.cur, n, g_flags + 2);
    else k_any_differs<3><<<blocks, 256>>>(g_img.cur, n, g_flags + 2);
    uint32_t f = 0;
    HIPCHK(hipMemcpy(&f, g_flags + 2, 4, hipMemcpyDeviceToHost));
    *single = f ? 0 : 1;
    return 0;
}
extern "C" int xpnghip_image_fetch(uint8_t *dst) {
    if (!g_img.open || !dst) return fail("no staged image");
    HIPCHK(hipSetDevice(base_device()));
    HIPCHK(hipMemcpy(dst, g_img.cur, g_img.w * g_img.h * (uint64_t)g_img.pxsz, hipMemcpyDeviceToHost));
    return 0;
}
static int image_encode_impl(uint64_t T, int mode, uint8_t **blobs, uint64_t *blobs_len) {
    if (!g_img.open || !blobs || !blobs_len) return fail("no staged image");
    HIPCHK(hipSetDevice(base_device()));
    HIPCHK(hipDeviceSynchronize());  // (staging ran on the null stream)
    g_call++;
    const uint64_t N = tile_count_for(g_img.w, g_img.h);
    const int D = devices_for(T, N);
    if (D > 1) return encode_multi(D, mode, nullptr, g_img.cur, g_img.w, g_img.h, g_img.pxsz, blobs, blobs_len);
    xpnghip_ctx *c = cached_ctx(base_device(), g_img.w, g_img.h, g_img.pxsz);
    if (!c) return 1;
    if (ensure_buf(c->d_blobs, c->cap_blobs, xpnghip_ctx_blob_bound(c, 0, N))) return 1;
    uint64_t len = 0;
    if (xpnghip_encode_device(c, mode, g_img.cur, 0, N, c->d_blobs, &len, nullptr)) return 1;
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    if (!out) return fail("malloc failed");
    if (hipMemcpy(out, c->d_blobs, len, hipMemcpyDeviceToHost) != hipSuccess) { free(out); return fail("blob download failed"); }
    *blobs = out; *blobs_len = len;
    return 0;
}
extern "C" int xpnghip_image_encode_T(uint64_t T, int mode, uint8_t **blobs, uint64_t *blobs_len) { XPNG_GUARDED(image_encode_impl(T, mode, blobs, blobs_len)) }
extern "C" int xpnghip_image_encode(int mode, uint8_t **blobs, uint64_t *blobs_len) { XPNG_GUARDED(image_encode_impl(1, mode, blobs, blobs_len)) }
// devices a call with worker count T would use on an image of this geometry (what xpng_store_T prints in its MPx/s line)
extern "C" int xpnghip_devices_for(uint64_t T, uint64_t w, uint64_t h) {
    if (!w || !h || w > (1u << 24) || h > (1u << 24)) return 0;
    return devices_for(T, tile_count_for(w, h));
}
