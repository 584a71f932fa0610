// wrappers.hpp -- the host-buffer entry points of include/xpng_hip.h (included by xpng_hip.hip): what the host C driver
// calls in place of the reference's two thread fan-outs (libxpng.c:758 + 764-769, 982-983).
//
//   * RE-ENTRANT, like the reference (SURVEY 8(b) "Threading": no globals, every call spawns and joins its own workers,
//     until_fork/4_letters.c:9-17).  A call owns everything it touches: it CHECKS OUT a context (tile table + workspace + stream)
//     and, for xpng_store, a staged-image object from a pool of idle ones, works on them without any lock, and hands them back.
//     The pool's mutex is held only while an object is taken out or put back - never across a HIP call - so two host threads
//     in xpng_store / xpng_load run side by side on two contexts and two streams.
//   * One process, up to T devices.  The reference's `T` is its worker count (libxpng.c:146-151: T = min(T, N) threads over a
//     shared tile cursor); here T devices each take one contiguous, pixel-weighted tile range (the same split as
//     xpng_amd/shard.py), encode / decode it from their own band of the raster, and the blob ranges are gathered on the first
//     device by peer copies over xGMI for the one concatenation (libxpng.c:764-769) before the single copy to the host.
//     T == 0 ("auto") means ONE device: the multi-device path has never run on real peers (SCALE skipped in every round so
//     far), so it is opt-in - T > 1, or XPNG_GPUS=n for T == 0 - until a byte-parity run on hardware exists.
//   * Idle contexts keep their staging buffers, which only ever grow: repeat calls allocate nothing.
//   * Everything runs on explicit devices (XPNG_DEVICE = first device, default 0) and the caller's current device is restored.
#pragma once

static int base_device() {
    static const int d = [] { const char *e = getenv("XPNG_DEVICE"); return e ? atoi(e) : 0; }();
    return d;
}
// (probe builds) XPNG_FAKE_DEVICES=n: rehearse the multi-device path on a box with one GPU (n shards, all on the base device)
static int fake_devices() {
    const char *e = probe_env("XPNG_FAKE_DEVICES");
    return e ? atoi(e) : 0;
}
static int usable_devices() {
    const int f = fake_devices();
    if (f > 0) return xpnghip_device_count() > base_device() ? f : 0;
    const int n = xpnghip_device_count() - base_device();
    return n > 0 ? n : 0;
}
static int shard_device(int k) { return fake_devices() > 0 ? base_device() : base_device() + k; }

// (probe builds) XPNG_TRACE_API=1: wall-clock marks of the host-buffer calls on stderr
struct ApiTrace {
    bool on; std::chrono::steady_clock::time_point t0;
    ApiTrace() : on(probe_env("XPNG_TRACE_API") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char *what, int k = -1) const {
        if (on) fprintf(stderr, "[xpng api] %8.3f ms  %s %d\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what, k);
    }
};

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

// ---- pool of idle contexts ------------------------------------------------------------------------------------
// g_pool_mu guards the three idle lists below and nothing else; no HIP call is made while it is held.
static std::mutex g_pool_mu;
static std::vector<xpnghip_ctx *> g_idle;  // most recently used first
constexpr size_t POOL_MAX = 12;
constexpr uint64_t POOL_MAX_BYTES = 64ull << 30;

// takes every idle context beyond `keep` entries / `keep_bytes` out of the pool; the caller destroys them (outside the lock)
static std::vector<xpnghip_ctx *> pool_trim_locked(size_t keep, uint64_t keep_bytes) {
    std::vector<xpnghip_ctx *> kept, victims;
    uint64_t sum = 0;
    for (xpnghip_ctx *c : g_idle) {
        sum += c->ws_bytes + c->cap_raster + c->cap_blobs + c->cap_blob_in;
        if (kept.size() < keep && sum <= keep_bytes) kept.push_back(c); else victims.push_back(c);
    }
    g_idle.swap(kept);
    return victims;
}
static xpnghip_ctx *ctx_checkout(int dev, uint64_t w, uint64_t h, int pxsz, uint64_t r0 = 0, uint64_t r1 = ~0ull) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_idle.size(); i++) {
            xpnghip_ctx *c = g_idle[i];
            if (c->device == dev && c->W == w && c->H == h && c->pxsz == pxsz && c->r0 == r0 && c->r1 == std::min<uint64_t>(r1, c->tiles.size())) {
                g_idle.erase(g_idle.begin() + (long)i);
                return c;
            }
        }
    }
    xpnghip_ctx *c = nullptr;
    if (xpnghip_ctx_create_range(&c, dev, w, h, pxsz, 1, r0, r1)) {
        // out of memory, perhaps: give the idle workspaces back and try once more
        std::vector<xpnghip_ctx *> victims;
        { std::lock_guard<std::mutex> lk(g_pool_mu); victims = pool_trim_locked(0, 0); }
        for (xpnghip_ctx *v : victims) xpnghip_ctx_destroy(v);
        if (xpnghip_ctx_create_range(&c, dev, w, h, pxsz, 1, r0, r1)) return nullptr;
    }
    return c;
}
static void ctx_checkin(xpnghip_ctx *c) {
    if (!c) return;
    std::vector<xpnghip_ctx *> victims;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_idle.insert(g_idle.begin(), c);
        victims = pool_trim_locked(POOL_MAX, POOL_MAX_BYTES);
    }
    for (xpnghip_ctx *v : victims) xpnghip_ctx_destroy(v);
}
// Nothing of the call that held a context may still be running when the context goes back to the pool: an early error exit (a
// failed launch or allocation for shard k > 0, a corrupt blob) leaves kernels and copies of the earlier shards in flight - copies
// that read or write the CALLER's buffers, which the caller may free as soon as the call has returned, and workspaces the next
// thread to check this context out would drive from another stream with nothing ordering the two.  On the success
// paths these streams are idle already and the synchronisations return at once.
static void ctx_quiesce(xpnghip_ctx *c) {
    if (!c) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(c->device) == hipSuccess) {
        hipStream_t qs[5] = {c->stream, c->enc_side, c->enc_side2, c->dec.side, c->dec.side2};
        for (hipStream_t q : qs) if (q) (void)hipStreamSynchronize(q);
    }
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
}
struct CtxLease {  // a context for the duration of one call
    xpnghip_ctx *c = nullptr;
    CtxLease() = default;
    explicit CtxLease(xpnghip_ctx *p) : c(p) {}
    CtxLease(const CtxLease &) = delete;
    CtxLease &operator=(const CtxLease &) = delete;
    CtxLease(CtxLease &&o) noexcept : c(o.c) { o.c = nullptr; }
    ~CtxLease() { ctx_quiesce(c); ctx_checkin(c); }
};

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
// how many devices a call uses: T >= 1 -> min(T, usable, N); T == 0 -> ONE (see the header comment), or XPNG_GPUS
static int devices_for(uint64_t T, uint64_t N) {
    const int have = usable_devices();
    if (have < 1) return 0;
    uint64_t d = T ? T : 1;
    if (const char *e = getenv("XPNG_GPUS")) if (!T && atoi(e) > 0) d = (uint64_t)atoi(e);
    d = std::min<uint64_t>(d, (uint64_t)have);
    d = std::min<uint64_t>(d, N);
    return (int)std::max<uint64_t>(d, 1);
}
// Contiguous ranges balanced by pixel count: range k ends at the first tile where the running pixel count reaches k/D of the
// total, and every range keeps at least one tile (xpng_amd/shard.py weighted_tile_ranges is the same rule; the reference's
// cursor hands out tiles one by one, libxpng.c:150-151 - any partition gives the same bytes).
static void shard_ranges(const std::vector<TileDesc> &tiles, int D, std::vector<std::pair<uint64_t, uint64_t>> &out) {
    out.clear();
    const uint64_t N = tiles.size();
    if (D < 1) D = 1;
    if ((uint64_t)D > N) D = (int)N;
    uint64_t total = 0, acc = 0, start = 0;
    for (const TileDesc &t : tiles) total += t.n;
    int k = 1;
    for (uint64_t i = 0; i < N && k < D; i++) {
        acc += tiles[i].n;
        const uint64_t left = N - (i + 1);  // tiles behind this one: each of the D - k later ranges needs one
        if (acc * (uint64_t)D >= total * (uint64_t)k || left == (uint64_t)(D - k)) {
            out.emplace_back(start, i + 1);
            start = i + 1; k++;
        }
    }
    out.emplace_back(start, N);
}
static std::vector<Shard> make_shards(const std::vector<TileDesc> &tiles, int D) {
    std::vector<std::pair<uint64_t, uint64_t>> rr;
    shard_ranges(tiles, D, rr);
    std::vector<Shard> out;
    for (size_t s = 0; s < rr.size(); s++) {
        Shard sh{shard_device((int)s), rr[s].first, rr[s].second, 0, 0, nullptr, 0, 0};
        uint32_t y0 = ~0u, y1 = 0;
        for (uint64_t i = sh.r0; i < sh.r1; i++) { y0 = std::min(y0, tiles[i].y); y1 = std::max(y1, tiles[i].y + tiles[i].h); }
        sh.y0 = y0; sh.y1 = y1;
        out.push_back(sh);
    }
    return out;
}
// ONE device, one ordinary image (the narrow regime: at most 2048 (tile, stream) pairs): the call is cut into tile-row groups
// that run on the SAME device, each on a context and stream of its own, so that transfers and kernels of one call overlap
// (6.1 of the 22.5 ms of a 4096^2 decode call were copies):
//   encode - group 0 is the first tile row: it holds the biggest tile, whose alpha chain is the critical path of the call, and
//            its band is the first 13 % of the raster, so that chain starts 0.4 ms into the call instead of 2.4;
//   decode - the groups without the biggest tile finish ~3 ms earlier and their bands (87 % of the raster) travel to the host
//            while the first row's chains are still running; bands of whole tile rows are contiguous: one 1-D copy each.
// The bytes are those of the unsplit call (tiles are coded independently; the same code path as T > 1 devices).
static bool pipeline_shards(const std::vector<TileDesc> &tiles, int pxsz, uint64_t raster_bytes, int dev, std::vector<Shard> &out) {
    out.clear();
    const uint64_t N = tiles.size();
    if (N * (pxsz == 4 ? 10 : 9) > 2048 || raster_bytes < (24u << 20) || probe_env("XPNG_NO_PIPELINE_SHARDS")) return false;
    std::vector<uint64_t> row_start;  // first tile of every tile row
    for (uint64_t i = 0; i < N; i++) if (i == 0 || tiles[i].y != tiles[i - 1].y) row_start.push_back(i);
    const uint64_t R = row_start.size();
    if (R < 3) return false;
    row_start.push_back(N);
    const uint64_t mid = 1 + (R - 1) / 2;  // rows [1, mid) and [mid, R)
    const uint64_t cuts[4] = {0, 1, mid, R};
    for (int k = 0; k < 3; k++) {
        Shard sh{dev, row_start[cuts[k]], row_start[cuts[k + 1]], 0, 0, nullptr, 0, 0};
        sh.y0 = tiles[sh.r0].y; sh.y1 = tiles[sh.r1 - 1].y + tiles[sh.r1 - 1].h;
        out.push_back(sh);
    }
    return true;
}

// host-only (no device needed): the tile ranges a call on D devices would use, for the CPU tests that cross-check this
// split against xpng_amd/shard.py; ranges[2k], ranges[2k+1] = [r0, r1) of device k.  Returns the number of ranges.
extern "C" int xpnghip_shard_ranges(uint64_t w, uint64_t h, int D, uint64_t *ranges, int cap) {
    try {
        if (!w || !h || w > (1u << 24) || h > (1u << 24) || D < 1 || !ranges) return -1;
        std::vector<TileDesc> tiles;
        build_tiles(w, h, tiles);
        std::vector<std::pair<uint64_t, uint64_t>> rr;
        shard_ranges(tiles, D, rr);
        if ((int)rr.size() > cap) return -1;
        for (size_t k = 0; k < rr.size(); k++) { ranges[2 * k] = rr[k].first; ranges[2 * k + 1] = rr[k].second; }
        return (int)rr.size();
    } catch (...) { return -1; }
}

// Peer access between the gather device and a shard's device: asked once per ordered pair.  hipMemcpyPeerAsync works either
// way; without peer access the runtime stages the copy through host memory, which the caller should know about: the note is
// left in xpnghip_last_error() although the call succeeds.
static bool peer_ready(int a, int b) {
    if (a == b) return true;
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, int>, bool>> known;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : known) if (e.first.first == a && e.first.second == b) return e.second;
    }
    int can = 0;
    bool ok = hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can;
    if (ok) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        ok = hipSetDevice(a) == hipSuccess;
        if (ok) {
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            (void)hipGetLastError();
        }
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    std::lock_guard<std::mutex> lk(mu);
    known.push_back({{a, b}, ok});
    return ok;
}
static void note_peers(const std::vector<Shard> &sh) {
    std::string note;
    for (size_t k = 1; k < sh.size(); k++)
        if (!(peer_ready(sh[0].dev, sh[k].dev) && peer_ready(sh[k].dev, sh[0].dev)))
            note += (note.empty() ? "note: no peer access between devices " : ", ") + std::to_string(sh[0].dev) + "<->" + std::to_string(sh[k].dev);
    if (!note.empty()) g_err = note + ": those copies were staged through host memory";
}

static int check_geometry(uint64_t w, uint64_t h, int pxsz) {
    if (!w || !h || w > (1u << 24) || h > (1u << 24) || (pxsz != 3 && pxsz != 4)) return fail("bad raster geometry");
    if (usable_devices() < 1) return fail("no usable HIP device (libxpng_hip has no CPU fallback)");
    return 0;
}

// Touches the caller's (typically freshly malloc()ed) raster, one access per page, on a few helper threads while the kernels
// run: its first-touch page faults (16 k of them for a 4096^2 RGBA raster: ~4 ms) otherwise land inside the download.  A page is
// faulted in by writing back the byte that was read from it: the contents do not change, so a file that is rejected later leaves
// the caller's buffer as it was, at every raster size (the decode contract: a rejected tile's pixels stay untouched).
// If a helper thread cannot be created the pages simply fault during the copy.
struct Prefault {
    static constexpr int NT = 4;
    std::thread th[NT];
    int started = 0;
    void start(uint8_t *raster, uint64_t s) {
        const uint64_t part = ((s / NT) + 4095) & ~4095ull;
        for (int t = 0; t < NT; t++) {
            const uint64_t a = std::min<uint64_t>(s, t * part), b = std::min<uint64_t>(s, (t + 1) * part);
            try { th[t] = std::thread([=] { volatile uint8_t *p = raster; for (uint64_t o = a; o < b; o += 4096) { const uint8_t v = p[o]; p[o] = v; } }); }
            catch (...) { return; }  // (std::system_error: no more threads - the ones already started are joined below)
            started = t + 1;
        }
    }
    void join() { for (int t = 0; t < started; t++) if (th[t].joinable()) th[t].join(); started = 0; }
    ~Prefault() { join(); }  // (a joinable std::thread must never be destroyed: that is std::terminate, past every catch)
};
// pinned staging -> the caller's pageable raster on a few helper threads (the runtime's own pageable download moves ~10 GB/s
// through one thread: 6.7 ms for a 4096^2 RGBA raster; device -> pinned runs at link speed and four copy threads at ~30 GB/s).
// When a thread cannot be created its part is copied by the caller's thread, at once.
struct CopyOut {
    std::vector<std::thread> th;
    void start(uint8_t *dst, const uint8_t *src, uint64_t n) {
        constexpr int NT = 4;
        const uint64_t part = ((n / NT) + 4095) & ~4095ull;
        for (int t = 0; t < NT; t++) {
            const uint64_t a = std::min<uint64_t>(n, t * part), b = std::min<uint64_t>(n, (t + 1) * part);
            if (a == b) continue;
            try { th.emplace_back([=] { memcpy(dst + a, src + a, b - a); }); }
            catch (...) { memcpy(dst + a, src + a, b - a); }
        }
    }
    void join() { for (std::thread &t : th) if (t.joinable()) t.join(); th.clear(); }
    ~CopyOut() { join(); }
};
static int ensure_stage(xpnghip_ctx *c, uint64_t need) {
    if (c->cap_stage >= need && c->h_stage) return 0;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->cap_stage = 0;
    HIPCHK(hipHostMalloc((void **)&c->h_stage, need));
    c->cap_stage = need;
    return 0;
}

// Encode on D devices.  The raster is either in host memory (h_src) or already staged on the base device (d_src; the caller
// has synchronised the stream that produced it).
static int encode_multi(std::vector<Shard> sh, const std::vector<TileDesc> &tiles, int mode, const uint8_t *h_src, const uint8_t *d_src, uint64_t w, uint64_t h, int pxsz, uint8_t **blobs, uint64_t *blobs_len) {
    const uint64_t bpr = w * (uint64_t)pxsz;
    const ApiTrace tr;
    std::vector<CtxLease> leases;
    leases.reserve(sh.size());
    const int dev0 = sh[0].dev;
    note_peers(sh);
    // shard 0 encodes straight into the gather buffer (sized for the whole image), the others into their own
    uint64_t whole_bound = 16;
    for (const TileDesc &t : tiles) whole_bound += (uint64_t)t.n * pxsz + 4;
    for (size_t k = 0; k < sh.size(); k++) {
        Shard &s = sh[k];
        HIPCHK(hipSetDevice(s.dev));
        if (!(s.c = ctx_checkout(s.dev, w, h, pxsz, s.r0, s.r1))) return 1;
        leases.emplace_back(s.c);
        xpnghip_ctx *c = s.c;
        const uint64_t band = (uint64_t)(s.y1 - s.y0) * bpr;
        if (ensure_buf(c->d_raster, c->cap_raster, band + 16)) return 1;
        if (ensure_buf(c->d_blobs, c->cap_blobs, k == 0 ? whole_bound : xpnghip_ctx_blob_bound(c, s.r0, s.r1))) return 1;
        // kernels address rows absolutely: the band is handed over as if the whole raster were there (only rows [y0, y1) are
        // touched); it starts at the 16-byte phase row y0 has in the whole raster, so the virtual base stays 16-byte aligned
        uint8_t *bandp = c->d_raster + (((uint64_t)s.y0 * bpr) & 15);
        if (h_src) HIPCHK(hipMemcpyAsync(bandp, h_src + (uint64_t)s.y0 * bpr, band, hipMemcpyHostToDevice, ctx_stream(c)));
        else if (s.dev == dev0) HIPCHK(hipMemcpyAsync(bandp, d_src + (uint64_t)s.y0 * bpr, band, hipMemcpyDeviceToDevice, ctx_stream(c)));
        else HIPCHK(hipMemcpyPeerAsync(bandp, s.dev, d_src + (uint64_t)s.y0 * bpr, dev0, band, ctx_stream(c)));
        if (xpnghip_encode_device(c, mode, bandp - (uint64_t)s.y0 * bpr, s.r0, s.r1, c->d_blobs, nullptr, nullptr)) return 1;
        tr.mark("encode launched, shard", (int)k);
    }
    uint64_t total = 0;
    for (Shard &s : sh) {
        HIPCHK(hipSetDevice(s.dev));
        HIPCHK(hipStreamSynchronize(ctx_stream(s.c)));
        s.len = s.c->h_total[0]; s.off = total; total += s.len;
        tr.mark("encode done, shard", (int)(&s - &sh[0]));
    }
    // the one exchange of the path (libxpng.c:764-769): blob ranges -> device 0, behind shard 0's own bytes
    for (size_t k = 1; k < sh.size(); k++) {
        Shard &s = sh[k];
        HIPCHK(hipSetDevice(s.dev));
        if (s.dev == dev0) HIPCHK(hipMemcpyAsync(sh[0].c->d_blobs + s.off, s.c->d_blobs, s.len, hipMemcpyDeviceToDevice, ctx_stream(s.c)));
        else HIPCHK(hipMemcpyPeerAsync(sh[0].c->d_blobs + s.off, dev0, s.c->d_blobs, s.dev, s.len, ctx_stream(s.c)));
    }
    for (size_t k = 1; k < sh.size(); k++) { HIPCHK(hipSetDevice(sh[k].dev)); HIPCHK(hipStreamSynchronize(ctx_stream(sh[k].c))); }
    // (a buffer allocated for the raw bound up front and touched by helper threads while the kernels run, plus a pinned staging
    //  copy, measured SLOWER - 14.2 against 11.5 ms for the call: the helper threads compete with the runtime's pageable upload of
    //  the later bands for the host's memory bandwidth, and the first shard's chains start late)
    uint8_t *out = (uint8_t *)malloc(total ? total : 1);
    if (!out) return fail("malloc failed");
    HIPCHK(hipSetDevice(dev0));
    // (on the shard's own stream: a plain hipMemcpy is a null-stream operation and waits for every blocking stream of the device,
    //  i.e. for the calls other host threads have in flight)
    if (hipMemcpyAsync(out, sh[0].c->d_blobs, total, hipMemcpyDeviceToHost, ctx_stream(sh[0].c)) != hipSuccess || hipStreamSynchronize(ctx_stream(sh[0].c)) != hipSuccess) { free(out); return fail("blob download failed"); }
    tr.mark("blobs on the host", 0);
    if (sh.back().dev != dev0) {
        // More than one REAL device: this path (peer copies into device 0's gather buffer) has been rehearsed with fake devices and
        // gloo ranks but has not yet run on real peers under this library's test-suite (tests/test_gpu_parity.py
        // test_two_real_devices_code_the_same_bytes skips on one-GPU boxes).  So the gathered bytes are checked before they are
        // handed out - the size words of the N tiles must walk exactly to the total (libxpng.c:764-769, 982) - and the call says so.
        uint64_t o = 0;
        for (size_t i = 0; i < tiles.size() && o + 4 <= total; i++) { uint32_t h0; memcpy(&h0, out + o, 4); const uint32_t sz = h0 & 0xFFFFFFu; if (!sz) { o = ~0ull; break; } o += sz; }
        if (o != total) { free(out); return fail("multi-device gather: the tile sizes do not walk to the gathered length (peer copies misplaced?)"); }
        if (g_err.rfind("note:", 0) != 0) g_err = "note: coded on " + std::to_string(sh.size()) + " devices; this multi-device form is opt-in (T > 1) and awaits a byte-parity run on real peer GPUs";
    }
    *blobs = out; *blobs_len = total;
    return 0;
}

static int encode_tiles_impl(uint64_t T, int mode, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz, uint8_t **blobs, uint64_t *blobs_len) {
    if (!raster || !blobs || !blobs_len) return fail("null argument");
    if (check_geometry(w, h, pxsz)) return 1;
    DevGuard guard;
    const uint64_t N = tile_count_for(w, h);
    const int D = devices_for(T, N);
    const uint64_t s = w * h * (uint64_t)pxsz;
    if (D > 1 || s >= (24u << 20)) {
        std::vector<TileDesc> tiles;
        build_tiles(w, h, tiles);
        std::vector<Shard> sh;
        if (D > 1) return encode_multi(make_shards(tiles, D), tiles, mode, raster, nullptr, w, h, pxsz, blobs, blobs_len);
        if (pipeline_shards(tiles, pxsz, s, base_device(), sh)) return encode_multi(sh, tiles, mode, raster, nullptr, w, h, pxsz, blobs, blobs_len);
    }
    HIPCHK(hipSetDevice(base_device()));
    CtxLease lease(ctx_checkout(base_device(), w, h, pxsz));
    xpnghip_ctx *c = lease.c;
    if (!c) return 1;
    if (ensure_buf(c->d_raster, c->cap_raster, s) || ensure_buf(c->d_blobs, c->cap_blobs, xpnghip_ctx_blob_bound(c, 0, N))) return 1;
    HIPCHK(hipMemcpyAsync(c->d_raster, raster, s, hipMemcpyHostToDevice, ctx_stream(c)));
    uint64_t len = 0;
    if (xpnghip_encode_device(c, mode, c->d_raster, 0, N, c->d_blobs, &len, nullptr)) return 1;
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    if (!out) return fail("malloc failed");
    if (hipMemcpyAsync(out, c->d_blobs, len, hipMemcpyDeviceToHost, ctx_stream(c)) != hipSuccess || hipStreamSynchronize(ctx_stream(c)) != hipSuccess) { free(out); return fail("blob download failed"); }
    *blobs = out; *blobs_len = len;
    return 0;
}

// Decode on D devices: the host walks the tile sizes (the file is in host memory, libxpng.c:982), every device gets the blob
// range of its tiles and fills its band; the bands come back as one rectangle per tile row of the range.
static int decode_multi(std::vector<Shard> sh, const std::vector<TileDesc> &tiles, int mode, const uint8_t *blobs, const std::vector<uint64_t> &off, uint64_t w, uint64_t h, int pxsz, uint8_t *raster,
                        bool first_last) {
    const uint64_t bpr = w * (uint64_t)pxsz;
    const ApiTrace tr;
    std::vector<CtxLease> leases;
    leases.reserve(sh.size());
    std::vector<uint64_t> rel;
    for (Shard &s : sh) {
        HIPCHK(hipSetDevice(s.dev));
        if (!(s.c = ctx_checkout(s.dev, w, h, pxsz, s.r0, s.r1))) return 1;
        leases.emplace_back(s.c);
        xpnghip_ctx *c = s.c;
        s.off = off[s.r0]; s.len = off[s.r1] - off[s.r0];
        const uint64_t band = (uint64_t)(s.y1 - s.y0) * bpr;
        if (ensure_buf(c->d_raster, c->cap_raster, band + 16) || ensure_buf(c->d_blob_in, c->cap_blob_in, s.len)) return 1;
        HIPCHK(hipMemcpyAsync(c->d_blob_in, blobs + s.off, s.len, hipMemcpyHostToDevice, ctx_stream(c)));
        rel.assign(off.begin() + (long)s.r0, off.begin() + (long)s.r1);
        for (uint64_t &o : rel) o -= s.off;
        uint8_t *bandp = c->d_raster + (((uint64_t)s.y0 * bpr) & 15);
        if (xpnghip_decode_device(c, mode, c->d_blob_in, s.len, rel.data(), s.r0, s.r1, bandp - (uint64_t)s.y0 * bpr, nullptr)) return 1;
        tr.mark("decode launched, shard", (int)(&s - &sh[0]));
    }
    Prefault pf;
    CopyOut out;
    const uint64_t rbytes = bpr * h;
    if (rbytes >= (32u << 20) && !probe_env("XPNG_NO_PREFAULT")) pf.start(raster, rbytes);
    int rc = 0;
    // bands come back in the order the shards finish: with pipeline shards the first one (the biggest tile's row) is the last
    for (size_t q = 0; q < sh.size(); q++) {
        Shard &s = sh[first_last ? (q + 1) % sh.size() : q];
        HIPCHK(hipSetDevice(s.dev));
        const int st = xpnghip_ctx_decode_status(s.c, nullptr);
        tr.mark("decode done, shard", (int)(&s - &sh[0]));
        pf.join();
        if (st == 1) rc = fail("corrupt file: a tile header is inconsistent with the tile table");
        else if (st != 0) rc = fail("decode failed");
        if (rc) continue;
        const bool whole_rows = tiles[s.r0].x == 0 && tiles[s.r1 - 1].x + tiles[s.r1 - 1].w == w;
        if (whole_rows) {  // a band of whole tile rows is contiguous in both rasters: one 1-D copy
            const uint64_t band = (uint64_t)(s.y1 - s.y0) * bpr;
            const uint8_t *d_band = s.c->d_raster + (((uint64_t)s.y0 * bpr) & 15);
            if (first_last && band >= (4u << 20) && !ensure_stage(s.c, band)) {
                // device -> pinned staging at link speed, then helper threads move it into the caller's raster while the next band
                // (or the last shard's chains) is still on the device
                if (hipMemcpyAsync(s.c->h_stage, d_band, band, hipMemcpyDeviceToHost, ctx_stream(s.c)) != hipSuccess || hipStreamSynchronize(ctx_stream(s.c)) != hipSuccess) { rc = fail("raster download failed"); continue; }
                tr.mark("band in pinned staging, shard", (int)(&s - &sh[0]));
                out.start(raster + (uint64_t)s.y0 * bpr, s.c->h_stage, band);
            } else if (hipMemcpyAsync(raster + (uint64_t)s.y0 * bpr, d_band, band, hipMemcpyDeviceToHost, ctx_stream(s.c)) != hipSuccess) rc = fail("raster download failed");
            continue;
        }
        for (uint64_t i = s.r0; i < s.r1;) {  // tiles i..j-1 share a tile row: one rectangle
            uint64_t j = i + 1;
            while (j < s.r1 && tiles[j].y == tiles[i].y) j++;
            const uint64_t x0 = tiles[i].x, x1 = tiles[j - 1].x + tiles[j - 1].w, y = tiles[i].y;
            if (hipMemcpy2DAsync(raster + y * bpr + x0 * pxsz, bpr, s.c->d_raster + (((uint64_t)s.y0 * bpr) & 15) + (y - s.y0) * bpr + x0 * pxsz, bpr, (x1 - x0) * pxsz, tiles[i].h,
                                 hipMemcpyDeviceToHost, ctx_stream(s.c)) != hipSuccess) { rc = fail("raster download failed"); break; }
            i = j;
        }
    }
    for (Shard &s : sh) { (void)hipSetDevice(s.dev); if (hipStreamSynchronize(ctx_stream(s.c)) != hipSuccess && !rc) rc = fail("raster download failed"); }
    out.join();  // (before the leases hand the contexts, and with them the staging buffers, back)
    tr.mark("raster complete", 0);
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
    DevGuard guard;
    const int D = devices_for(T, N);
    const uint64_t s = w * h * (uint64_t)pxsz;
    if (D > 1 || s >= (24u << 20)) {
        std::vector<TileDesc> tiles;
        build_tiles(w, h, tiles);
        std::vector<Shard> sh;
        if (D > 1) return decode_multi(make_shards(tiles, D), tiles, mode, blobs, off, w, h, pxsz, raster, false);
        if (pipeline_shards(tiles, pxsz, s, base_device(), sh)) return decode_multi(sh, tiles, mode, blobs, off, w, h, pxsz, raster, true);
    }
    HIPCHK(hipSetDevice(base_device()));
    CtxLease lease(ctx_checkout(base_device(), w, h, pxsz));
    xpnghip_ctx *c = lease.c;
    if (!c) return 1;
    if (ensure_buf(c->d_raster, c->cap_raster, s) || ensure_buf(c->d_blob_in, c->cap_blob_in, blobs_len)) return 1;
    HIPCHK(hipMemcpyAsync(c->d_blob_in, blobs, blobs_len, hipMemcpyHostToDevice, ctx_stream(c)));
    if (xpnghip_decode_device(c, mode, c->d_blob_in, blobs_len, off.data(), 0, N, c->d_raster, nullptr)) return 1;
    Prefault pf;
    if (s >= (32u << 20) && !probe_env("XPNG_NO_PREFAULT")) pf.start(raster, s);
    const int st = xpnghip_ctx_decode_status(c, nullptr);
    pf.join();
    if (st == 1) return fail("corrupt file: a tile header is inconsistent with the tile table");
    if (st != 0) return fail("decode failed");
    // (a pinned staging buffer with chunked copies and host copy threads measured no better than this plain copy: 21.8 against 21.5 ms)
    HIPCHK(hipMemcpyAsync(raster, c->d_raster, s, hipMemcpyDeviceToHost, ctx_stream(c)));
    HIPCHK(hipStreamSynchronize(ctx_stream(c)));
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

// ---- normalize_RGBA and the single-colour test on the device ---------------------------------------------------
// 16-byte flag blocks for the OR-reductions, per device: checked out for a call and put back, never freed per call (hipFree
// synchronises the whole device: a pipelined caller of xpnghip_normalize_device would stall all its streams on every call).
static std::vector<std::pair<int, uint32_t *>> g_idle_flags;  // (device, block); guarded by g_pool_mu
static uint32_t *flags_checkout(int dev) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_idle_flags.size(); i++)
            if (g_idle_flags[i].first == dev) { uint32_t *p = g_idle_flags[i].second; g_idle_flags.erase(g_idle_flags.begin() + (long)i); return p; }
    }
    uint32_t *p = nullptr;
    if (hipMalloc((void **)&p, 16) != hipSuccess) return nullptr;
    return p;
}
static void flags_checkin(int dev, uint32_t *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_idle_flags.push_back({dev, p});
}

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
static int normalize_device_impl(const void *d_rgba, uint64_t npx, void *d_out, int *pxsz_out, int *rewritten, void *stream) {
    if (!d_rgba || !d_out || !pxsz_out || !rewritten || !npx) return fail("null argument");
    if (((uintptr_t)d_rgba & 15) || ((uintptr_t)d_out & 3)) return fail("device buffers must be 16-byte aligned");
    int dev = 0;  // (the caller's raster lives on the caller's current device: so does the flag block)
    HIPCHK(hipGetDevice(&dev));
    uint32_t *flags = flags_checkout(dev);
    if (!flags) return fail("hipMalloc failed (flag block)");
    // (norm_device has read the flags back - a stream synchronisation - before it returns: the block is idle again)
    const int rc = norm_device(d_rgba, npx, d_out, pxsz_out, rewritten, (hipStream_t)stream, flags);
    flags_checkin(dev, flags);
    return rc;
}
extern "C" int xpnghip_normalize_device(const void *d_rgba, uint64_t npx, void *d_out, int *pxsz_out, int *rewritten, void *stream) {
    XPNG_GUARDED(normalize_device_impl(d_rgba, npx, d_out, pxsz_out, rewritten, stream))
}

// ---- staged image ------------------------------------------------------------------------------------------------
// One object per xpng_store call in flight (include/xpng_hip.h): upload buffer, rewrite buffer, flag block and streams of its
// own.  Idle objects are pooled with their buffers.
//
// Band pipeline [r4] (one device, an ordinary image: the geometry pipeline_shards accepts).  Through round 3 xpng_store uploaded the
// WHOLE raster, normalised it, read the two flags back and only then started ONE unsplit encode: the tile-row overlap that took
// xpnghip_encode_tiles from 13.7 to 12.1 ms never ran for the call the boundary replaces.  Now the
// raster moves in the same three tile-row bands, each on a stream of its own: upload -> k_norm_flags -> k_norm_zero_hidden_if (in
// place) -> the band's tile encode.  What makes that legal is the shape of normalize_RGBA (libxpng.c:688-721): the rewrite of
// hidden colours is per pixel, so a band can apply it from its own flag; the only whole-image question is "RGBA or RGB", and RGB
// needs NO pixel with alpha != 255 anywhere (a hidden pixel has alpha 0, so it is translucent too) - the first band that holds a
// translucent pixel settles it, and for an image with real alpha that is the first band, 13 % of the raster.  xpnghip_image_begin
// returns as soon as it is settled; the remaining bands are uploaded by whoever needs them next (the encode, band by band; a fetch
// or the single-colour test, all at once), which is why the caller's raster must stay valid until xpnghip_image_end.  An opaque
// RGBA raster is uploaded whole and repacked to RGB as before.
struct ImgBand { uint32_t y0, y1; bool issued; };
struct xpnghip_image {
    int dev = 0;
    uint64_t w = 0, h = 0, cap_in = 0, cap_norm = 0;
    int pxsz = 0, pxsz_in = 0;
    uint8_t *d_in = nullptr, *d_norm = nullptr;  // uploaded raster (hidden colours zeroed in place); repacked RGB raster
    const uint8_t *cur = nullptr;                // the staged (normalised) raster
    uint32_t *flags = nullptr;                   // device: [0..1] whole-image normalisation flags, [2] single colour, [4 + 2k ..] band k
    uint32_t *h_flags = nullptr;                 // pinned: band k's two flags at [2k]
    hipStream_t stream = nullptr;
    const uint8_t *h_src = nullptr;              // the caller's raster, while bands may still have to be uploaded
    std::vector<ImgBand> bands;                  // empty: whole-image form (everything is on `stream`)
    hipStream_t bs[3] = {nullptr, nullptr, nullptr};
};
static std::vector<xpnghip_image *> g_idle_images;  // guarded by g_pool_mu
constexpr size_t IMAGE_POOL_MAX = 4;
static void image_quiesce(xpnghip_image *im) {
    for (hipStream_t q : im->bs) if (q) (void)hipStreamSynchronize(q);
    if (im->stream) (void)hipStreamSynchronize(im->stream);
}
static void image_destroy(xpnghip_image *im) {
    if (!im) return;
    (void)hipSetDevice(im->dev);
    image_quiesce(im);
    for (hipStream_t q : im->bs) if (q) (void)hipStreamDestroy(q);
    if (im->stream) (void)hipStreamDestroy(im->stream);
    if (im->d_in) (void)hipFree(im->d_in);
    if (im->d_norm) (void)hipFree(im->d_norm);
    if (im->flags) (void)hipFree(im->flags);
    if (im->h_flags) (void)hipHostFree(im->h_flags);
    delete im;
}
static xpnghip_image *image_checkout(int dev) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_idle_images.size(); i++)
            if (g_idle_images[i]->dev == dev) { xpnghip_image *im = g_idle_images[i]; g_idle_images.erase(g_idle_images.begin() + (long)i); return im; }
    }
    xpnghip_image *im = new xpnghip_image();
    im->dev = dev;
    if (hipStreamCreateWithFlags(&im->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&im->flags, 64) != hipSuccess ||
        hipHostMalloc((void **)&im->h_flags, 64) != hipSuccess) { image_destroy(im); return nullptr; }
    return im;
}
static void image_checkin(xpnghip_image *im) {
    xpnghip_image *victim = nullptr;
    im->bands.clear(); im->h_src = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_idle_images.insert(g_idle_images.begin(), im);
        if (g_idle_images.size() > IMAGE_POOL_MAX) { victim = g_idle_images.back(); g_idle_images.pop_back(); }
    }
    image_destroy(victim);
}

// band k of the caller's raster -> d_in on the band's stream, with its share of normalize_RGBA behind it (launches only)
static int image_issue_band(xpnghip_image *im, size_t k) {
    ImgBand &b = im->bands[k];
    if (b.issued) return 0;
    const uint64_t bpr = im->w * (uint64_t)im->pxsz_in, off = (uint64_t)b.y0 * bpr, bytes = (uint64_t)(b.y1 - b.y0) * bpr;
    hipStream_t q = im->bs[k];
    HIPCHK(hipMemcpyAsync(im->d_in + off, im->h_src + off, bytes, hipMemcpyHostToDevice, q));
    if (im->pxsz_in == 4) {
        uint32_t *f = im->flags + 4 + 2 * k;
        const uint64_t npx = bytes / 4;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((npx / 4 + 255) / 256 + 1, 256 * 16);
        HIPCHK(hipMemsetAsync(f, 0, 8, q));
        k_norm_flags<<<blocks, 256, 0, q>>>((const uint32_t *)(im->d_in + off), npx, f);
        k_norm_zero_hidden_if<<<blocks, 256, 0, q>>>(f, (uint32_t *)(im->d_in + off), npx);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(im->h_flags + 2 * k, f, 8, hipMemcpyDeviceToHost, q));
    }
    b.issued = true;
    return 0;
}
// every band uploaded, normalised and finished: the staged raster is complete (whole-raster consumers: fetch, single colour,
// the multi-device encode, the repack to RGB)
static int image_complete(xpnghip_image *im) {
    for (size_t k = 0; k < im->bands.size(); k++) if (image_issue_band(im, k)) return 1;
    for (size_t k = 0; k < im->bands.size(); k++) HIPCHK(hipStreamSynchronize(im->bs[k]));
    return 0;
}

static int image_begin_impl(xpnghip_image **out, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz_in, int *pxsz_out) {
    if (!out || !raster || !pxsz_out) return fail("bad argument");
    *out = nullptr;
    if (check_geometry(w, h, pxsz_in)) return 1;
    DevGuard guard;
    HIPCHK(hipSetDevice(base_device()));
    xpnghip_image *im = image_checkout(base_device());
    if (!im) return fail("staging object allocation failed");
    auto body = [&]() -> int {
        const uint64_t s = w * h * (uint64_t)pxsz_in;
        if (ensure_buf(im->d_in, im->cap_in, s)) return 1;
        im->w = w; im->h = h; im->pxsz = im->pxsz_in = pxsz_in; im->cur = im->d_in;
        im->bands.clear(); im->h_src = raster;
        std::vector<TileDesc> tiles;
        std::vector<Shard> sh;
        if (s >= (24u << 20)) build_tiles(w, h, tiles);
        if (!tiles.empty() && pipeline_shards(tiles, pxsz_in, s, im->dev, sh) && sh.size() == 3 && !probe_env("XPNG_NO_IMAGE_BANDS")) {
            for (size_t k = 0; k < 3; k++)
                if (!im->bs[k]) HIPCHK(hipStreamCreateWithFlags(&im->bs[k], hipStreamNonBlocking));
            for (const Shard &q : sh) im->bands.push_back(ImgBand{q.y0, q.y1, false});
            if (pxsz_in == 3) return image_issue_band(im, 0);  // RGB: nothing to decide; the first band's upload starts now
            if (ensure_buf(im->d_norm, im->cap_norm, s)) return 1;
            for (size_t k = 0; k < 3; k++) {  // band by band until a translucent pixel settles "stays RGBA"
                if (image_issue_band(im, k)) return 1;
                HIPCHK(hipStreamSynchronize(im->bs[k]));
                if (im->h_flags[2 * k + 1]) return 0;
            }
            // no pixel with alpha != 255 anywhere: repack to RGB (libxpng.c:708-718); the whole raster is on the device by now
            const uint64_t npx = w * h;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((npx / 4 + 255) / 256 + 1, 256 * 16);
            k_norm_to_rgb<<<blocks, 256, 0, im->stream>>>((const uint32_t *)im->d_in, im->d_norm, npx);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(im->stream));
            im->cur = im->d_norm; im->pxsz = 3;
            return 0;
        }
        HIPCHK(hipMemcpyAsync(im->d_in, raster, s, hipMemcpyHostToDevice, im->stream));
        if (pxsz_in == 4) {
            if (ensure_buf(im->d_norm, im->cap_norm, s)) return 1;
            int rewritten = 0;
            if (norm_device(im->d_in, w * h, im->d_norm, &im->pxsz, &rewritten, im->stream, im->flags)) return 1;
            if (rewritten) im->cur = im->d_norm;
        }
        return 0;
    };
    if (body()) { image_quiesce(im); image_checkin(im); return 1; }
    *pxsz_out = im->pxsz;
    *out = im;
    return 0;
}
extern "C" int xpnghip_image_begin(xpnghip_image **img, const uint8_t *raster, uint64_t w, uint64_t h, int pxsz_in, int *pxsz_out) {
    XPNG_GUARDED(image_begin_impl(img, raster, w, h, pxsz_in, pxsz_out))
}
extern "C" void xpnghip_image_end(xpnghip_image *im) {
    if (!im) return;
    try {
        DevGuard guard;
        (void)hipSetDevice(im->dev);
        image_quiesce(im);  // (nothing of this call may still read the caller's raster or write the staging buffers)
        image_checkin(im);
    } catch (...) {}
}
static int image_single_colour_impl(xpnghip_image *im, int *single) {
    if (!im || !single) return fail("no staged image");
    DevGuard guard;
    HIPCHK(hipSetDevice(im->dev));
    if (image_complete(im)) return 1;
    const uint64_t n = im->w * im->h;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 16);
    HIPCHK(hipMemsetAsync(im->flags + 2, 0, 4, im->stream));
    if (im->pxsz == 4) k_any_differs<4><<<blocks, 256, 0, im->stream>>>(im->cur, n, im->flags + 2);
    else k_any_differs<3><<<blocks, 256, 0, im->stream>>>(im->cur, n, im->flags + 2);
    uint32_t f = 0;
    HIPCHK(hipMemcpyAsync(&f, im->flags + 2, 4, hipMemcpyDeviceToHost, im->stream));
    HIPCHK(hipStreamSynchronize(im->stream));
    *single = f ? 0 : 1;
    return 0;
}
extern "C" int xpnghip_image_single_colour(xpnghip_image *im, int *single) { XPNG_GUARDED(image_single_colour_impl(im, single)) }
static int image_fetch_impl(xpnghip_image *im, uint8_t *dst) {
    if (!im || !dst) return fail("no staged image");
    DevGuard guard;
    HIPCHK(hipSetDevice(im->dev));
    if (image_complete(im)) return 1;
    HIPCHK(hipMemcpyAsync(dst, im->cur, im->w * im->h * (uint64_t)im->pxsz, hipMemcpyDeviceToHost, im->stream));
    HIPCHK(hipStreamSynchronize(im->stream));
    return 0;
}
extern "C" int xpnghip_image_fetch(xpnghip_image *im, uint8_t *dst) { XPNG_GUARDED(image_fetch_impl(im, dst)) }

// the banded encode: tile-row group k is coded on band k's stream, right behind the band's upload and normalisation
static int image_encode_banded(xpnghip_image *im, std::vector<Shard> &sh, const std::vector<TileDesc> &tiles, int mode, uint8_t **blobs, uint64_t *blobs_len) {
    const ApiTrace tr;
    std::vector<CtxLease> leases;
    leases.reserve(sh.size());
    struct Quiesce { xpnghip_image *im; ~Quiesce() { for (hipStream_t q : im->bs) if (q) (void)hipStreamSynchronize(q); } } quiesce{im};  // (runs before the leases hand the contexts back, on every exit)
    uint64_t whole_bound = 16;
    for (const TileDesc &t : tiles) whole_bound += (uint64_t)t.n * im->pxsz + 4;
    for (size_t k = 0; k < sh.size(); k++) {
        Shard &s = sh[k];
        if (image_issue_band(im, k)) return 1;
        if (!(s.c = ctx_checkout(im->dev, im->w, im->h, im->pxsz, s.r0, s.r1))) return 1;
        leases.emplace_back(s.c);
        if (ensure_buf(s.c->d_blobs, s.c->cap_blobs, k == 0 ? whole_bound : xpnghip_ctx_blob_bound(s.c, s.r0, s.r1))) return 1;
        // (kernels address rows absolutely and touch only the rows of their tiles: the staged raster is handed over whole)
        if (xpnghip_encode_device(s.c, mode, im->cur, s.r0, s.r1, s.c->d_blobs, nullptr, im->bs[k])) return 1;
        tr.mark("encode launched, band", (int)k);
    }
    uint64_t total = 0;
    for (size_t k = 0; k < sh.size(); k++) {
        HIPCHK(hipStreamSynchronize(im->bs[k]));
        sh[k].len = sh[k].c->h_total[0]; sh[k].off = total; total += sh[k].len;
        tr.mark("encode done, band", (int)k);
    }
    for (size_t k = 1; k < sh.size(); k++) HIPCHK(hipMemcpyAsync(sh[0].c->d_blobs + sh[k].off, sh[k].c->d_blobs, sh[k].len, hipMemcpyDeviceToDevice, im->bs[k]));
    for (size_t k = 1; k < sh.size(); k++) HIPCHK(hipStreamSynchronize(im->bs[k]));
    uint8_t *out = (uint8_t *)malloc(total ? total : 1);
    if (!out) return fail("malloc failed");
    if (hipMemcpyAsync(out, sh[0].c->d_blobs, total, hipMemcpyDeviceToHost, im->bs[0]) != hipSuccess || hipStreamSynchronize(im->bs[0]) != hipSuccess) { free(out); return fail("blob download failed"); }
    tr.mark("blobs on the host", 0);
    *blobs = out; *blobs_len = total;
    return 0;
}
static int image_encode_impl(xpnghip_image *im, uint64_t T, int mode, uint8_t **blobs, uint64_t *blobs_len) {
    if (!im || !blobs || !blobs_len) return fail("no staged image");
    DevGuard guard;
    HIPCHK(hipSetDevice(im->dev));
    const uint64_t N = tile_count_for(im->w, im->h);
    const int D = devices_for(T, N);
    if (D > 1) {
        if (image_complete(im)) return 1;
        HIPCHK(hipStreamSynchronize(im->stream));  // (the shards' streams read the staged raster)
        std::vector<TileDesc> tiles;
        build_tiles(im->w, im->h, tiles);
        return encode_multi(make_shards(tiles, D), tiles, mode, nullptr, im->cur, im->w, im->h, im->pxsz, blobs, blobs_len);
    }
    if (!im->bands.empty()) {
        // the tile-row groups of the encode are the bands of the upload whenever the staged pixel format is the uploaded one (the
        // tile grid depends on w and h only); after a repack to RGB the raster is complete and the groups are cut afresh
        std::vector<TileDesc> tiles;
        std::vector<Shard> sh;
        build_tiles(im->w, im->h, tiles);
        bool same = pipeline_shards(tiles, im->pxsz, im->w * im->h * (uint64_t)im->pxsz, im->dev, sh) && sh.size() == im->bands.size();
        if (same && im->pxsz != im->pxsz_in) { if (image_complete(im)) return 1; }
        for (size_t k = 0; same && k < sh.size(); k++) same = sh[k].y0 == im->bands[k].y0 && sh[k].y1 == im->bands[k].y1;
        if (same) return image_encode_banded(im, sh, tiles, mode, blobs, blobs_len);
        if (image_complete(im)) return 1;
    }
    CtxLease lease(ctx_checkout(im->dev, im->w, im->h, im->pxsz));
    xpnghip_ctx *c = lease.c;
    if (!c) return 1;
    if (ensure_buf(c->d_blobs, c->cap_blobs, xpnghip_ctx_blob_bound(c, 0, N))) return 1;
    uint64_t len = 0;
    // on the image's stream: ordered behind the upload and the normalisation without a device-wide synchronisation
    if (xpnghip_encode_device(c, mode, im->cur, 0, N, c->d_blobs, &len, im->stream)) { (void)hipStreamSynchronize(im->stream); return 1; }
    uint8_t *out = (uint8_t *)malloc(len ? len : 1);
    if (!out) return fail("malloc failed");
    if (hipMemcpyAsync(out, c->d_blobs, len, hipMemcpyDeviceToHost, im->stream) != hipSuccess || hipStreamSynchronize(im->stream) != hipSuccess) { free(out); return fail("blob download failed"); }
    *blobs = out; *blobs_len = len;
    return 0;
}
extern "C" int xpnghip_image_encode_T(xpnghip_image *im, uint64_t T, int mode, uint8_t **blobs, uint64_t *blobs_len) { XPNG_GUARDED(image_encode_impl(im, T, mode, blobs, blobs_len)) }
extern "C" int xpnghip_image_encode(xpnghip_image *im, int mode, uint8_t **blobs, uint64_t *blobs_len) { XPNG_GUARDED(image_encode_impl(im, 1, mode, blobs, blobs_len)) }
// devices a call with worker count T would use on an image of this geometry (what xpng_store_T prints in its MPx/s line)
extern "C" int xpnghip_devices_for(uint64_t T, uint64_t w, uint64_t h) {
    if (!w || !h || w > (1u << 24) || h > (1u << 24)) return 0;
    return devices_for(T, tile_count_for(w, h));
}
// Gives every pooled object (idle contexts with their workspaces, staging objects, flag blocks) back to the runtime.  Optional:
// a process may simply exit.  Must not run concurrently with other calls into this library.
extern "C" void xpnghip_shutdown(void) {
    try {
        DevGuard guard;
        std::vector<xpnghip_ctx *> cs;
        std::vector<xpnghip_image *> ims;
        std::vector<std::pair<int, uint32_t *>> fl;
        { std::lock_guard<std::mutex> lk(g_pool_mu); cs.swap(g_idle); ims.swap(g_idle_images); fl.swap(g_idle_flags); }
        for (xpnghip_ctx *c : cs) xpnghip_ctx_destroy(c);
        for (xpnghip_image *im : ims) image_destroy(im);
        for (auto &f : fl) { (void)hipSetDevice(f.first); (void)hipFree(f.second); }
    } catch (...) {}
}
