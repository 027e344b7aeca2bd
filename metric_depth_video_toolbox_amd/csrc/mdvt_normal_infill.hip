// mdvt_normal_infill.hip -- basic_nomal_infill.normal_infill (basic_nomal_infill.py:87-119, "bni"): the step that consumes the
// stereo pair and its infill mask (movie_2_3D.py step 6).  The reference runs seven full-frame passes
// (masked_blur, infill_using_normals, cv2.blur, mark_lower_side, binary_dilation, two filter2D for blur_under_mask); what the
// result depends on is sparse -- the holes (3-5 % of a frame), a 4 x 4 fringe around them and a 13-pixel diamond around the
// holes' "lower side" -- so only those pixels are worked out here, each from the reference's own arithmetic:
//
//   bg(p)      = every channel of the mask image non-zero                                       bni:88
//   work       = img with bg pixels black                                                       bni:91
//   blur(q)    = masked_blur(work)(q): 6 x 6 Gaussian over the non-black pixels, 0 at a black q     bni:98, sr:114-153
//   filled(q)  = bg(q) ? blur(hit(q)) or 0 : blur(q), hit = the march of infill_using_normals        bni:101, sr:155-240
//                along ((mask/255)*2-1).xy through bg, preferring the sample two / one steps beyond the first non-bg one
//   work(p)    = box4(filled)(p) for bg pixels: cv2.blur's 4 x 4 mean, REFLECT_101, cvRound    bni:104-107
//   grown      = L1 distance <= 6 from a lower-side mark (six passes of the 4-neighbour cross)  bni:111-115
//   out(p)     = grown(p) ? 6 x 6 Gaussian of work over the grown pixels : work(p)              bni:118, 46-85
//
// blur() is evaluated only where somebody asks for it: at the 16 taps of a bg pixel's box window (marked in `need` by the bg
// pixels themselves) and at the hit of a marching bg pixel.  The tests hold it, bit for bit, to the seven passes written out
// in plain C (and those to the reference's own function, tests/golden/normal_infill.npz).
//
// Launches per set of images (about 16 B/px of workspace: two lists, four planes, the filled image).  The expensive parts -- the two marches and the Gaussians -- belong to
// a few per cent of the pixels, which sit together (holes are compact): done where they lie, a few workgroups would carry all
// of it (measured: 830 us per 8 images, 560 of them the marches on a handful of compute units).  So the dense passes only
// LIST pixels, and the work is dealt one listed pixel per lane over the whole chip:
//   k_ni_prep        dense, 4 px per thread as dwords: out := work image, the bg plane, list A = pixels with a non-black mask
//   k_ni_run<0>      list A: the box-window requests of the bg pixels (need plane); the lower-side march -> marks plane
//   k_ni_collect<2>  marks plane -> list M (every mark once: the rays of a hole side end on few pixels, and atomics on one word
//                    to find out who came first cost the kernel its tail)
//   k_ni_run<5>      list M: a mark sets its diamond of L1 radius 6 in the grown plane (+ the coarse per-tile map)
//   k_ni_collect<0>  need plane -> list B
//   k_ni_run<1>      list B: filled(q)
//   k_ni_run<2>      list A: the box mean for the bg pixels
//   k_ni_collect<1>  grown plane (tiles the coarse map names) -> list C
//   k_ni_run<3>      list C: blur_under_mask into the side buffer (the image is still read by the neighbours' taps)
//   k_ni_run<4>      list C: side buffer -> image
// The output image itself is the work image (the caller's input is never written).
#include "mdvt_device.h"

#include <stdio.h>
#include <stdlib.h>

#include <vector>

namespace mdvt {

namespace {

constexpr int kTileW = 128, kTileH = 16;      // pixels of a workgroup's tile in the collecting passes (and of the coarse map)
constexpr int kGrowR = 6;                     // bni:112: iterations of the cross dilation = L1 radius
constexpr int kNSub = 64;                     // sub-lists per list
constexpr int kHoleBatch = 8;                 // the same for the march through a hole (up to 400 steps)
constexpr int kMarchBatch = 8;                // samples of a march fetched together: one memory round trip per batch instead of per step

struct NiArgs {
    ImageSet img, mask, out;          // caller's images (u8 RGB rows); out doubles as the work image
    uint8_t* filled;                  // [n][H*W*3] valid where need != 0; later the side buffer of the last stage
    uint8_t* bg;                      // [n][H*W]
    uint8_t* need;                    // [n][H*W]
    uint8_t* grown;                   // [n][H*W] within six cross dilations of a lower-side mark (bni:111-115)
    uint8_t* marks;                   // [n][H*W] lower-side marks (bni:111-112)
    uint8_t* coarse;                  // [n][2][tiles_y][tiles_x] the tile holds a mark / may hold grown pixels
    // Lists.  One counter for a whole list serialises the appends (the atomics of one address take ~4 ns each, whoever sends
    // them: 0.7 ms per 8 images); so a list is kNSub sub-lists, a producing workgroup appends -- once, after compacting its
    // pixels in LDS -- to sub-list (its index % kNSub), whose segment has room for everything its producers could send.
    uint32_t* count;                  // [n][4][kNSub] entries of list A / B / C / M
    uint32_t* list_a;                 // [n][kNSub][cap_a] pixels with a non-black mask
    uint32_t* list_b;                 // [n][kNSub][cap_b] the marks (list M), then the requested pixels (B), then the grown pixels (C)
    uint32_t cap_a, cap_b;
    int W, H, tiles_x, tiles_y;
    int debug_skip;                   // ablation hook (env MDVT_NI_SKIP): 1 lower-side marches, 2 requests, 4 hole marches, 8 Gaussian taps
    BlurKernel K;
};

__device__ __forceinline__ int reflect101(int p, int len)       // cv::borderInterpolate(BORDER_REFLECT_101)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__device__ __forceinline__ bool ni_in_image(float x, float y, int W, int H)
{
    return x >= 0.0f && x < (float)W && y >= 0.0f && y < (float)H;
}

// Sample t of a march from (x, y) along (dx, dy): np.rint(p + d * t) in f32 (sr:205-207, ic:25-27), as integers -- the march is
// instruction bound, and "in the image?" on integers plus a 32-bit offset (rows x pitch and pixel strides stay below 2^24:
// checked by the launchers) is half the instructions of float compares and 64-bit addresses.  A sample outside the image
// gives `self` (never used).
__device__ __forceinline__ bool ni_sample(int x, int y, float dx, float dy, float t, int W, int H, uint32_t pitch, uint32_t px_bytes,
                                          uint32_t self, uint32_t& off)
{
    const int xi = (int)rintf((float)x + dx * t), yi = (int)rintf((float)y + dy * t);
    const bool in = (uint32_t)xi < (uint32_t)W && (uint32_t)yi < (uint32_t)H;
    off = in ? __umul24((uint32_t)yi, pitch) + __umul24((uint32_t)xi, px_bytes) : self;
    return in;
}

// A wave puts its flagged pixels into the workgroup's LDS list IN LANE ORDER (one LDS atomic per wave): listed neighbours
// stay neighbours, so that the lanes which later work on 64 consecutive entries read the same cache lines (with entries in
// the order the lanes' own atomics happened to land, every lane of a load had its own line: 54 line accesses per instruction).
__device__ __forceinline__ void ni_wave_put(uint32_t* lds_list, uint32_t* lds_cnt, bool flag, uint32_t value)
{
    const u64 m = __ballot(flag);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = (int)__ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(lds_cnt, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (flag) lds_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// The workgroup's compacted pixels (lds_list[0 .. lds_cnt)) go to its sub-list: one global atomic per workgroup.
__device__ __forceinline__ void ni_emit(uint32_t* sublist, uint32_t* counter, const uint32_t* lds_list, const uint32_t* lds_cnt, uint32_t* lds_base)
{
    __syncthreads();
    const uint32_t n = *lds_cnt;
    if (n == 0u) return;                                           // (workgroup-uniform)
    if (threadIdx.x == 0) *lds_base = atomicAdd(counter, n);
    __syncthreads();
    const uint32_t base = *lds_base;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) sublist[base + i] = lds_list[i];
}

// bni:88-91; list A
template <int PX>
__global__ void __launch_bounds__(128) k_ni_prep(NiArgs a)
{
    __shared__ uint32_t lds_list[128 * PX];
    __shared__ uint32_t lds_cnt, lds_base;
    const int g = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    const int W = a.W, H = a.H, x0 = g * PX;
    const size_t ib = (size_t)im * H * W;
    if (threadIdx.x == 0) lds_cnt = 0u;
    __syncthreads();
    uint32_t m[PX], c[PX];
#pragma unroll
    for (int q = 0; q < PX; ++q) { m[q] = 0u; c[q] = 0u; }
    const bool in = x0 < W;
    if (in) {
        const uint8_t* mrow = a.mask.image(im) + (size_t)y * a.mask.pitch;
        const uint8_t* irow = a.img.image(im) + (size_t)y * a.img.pitch;
        if (PX == 4) {
            const uint32_t* mp = (const uint32_t*)mrow + 3 * (size_t)g;
            const uint32_t* ip = (const uint32_t*)irow + 3 * (size_t)g;
            uint32_t mm[4], cc[4];
            unpack4(mp[0], mp[1], mp[2], mm);
            unpack4(ip[0], ip[1], ip[2], cc);
#pragma unroll
            for (int q = 0; q < PX; ++q) { m[q] = mm[q]; c[q] = cc[q]; }
        } else {
            m[0] = load_px_bytes(mrow, x0);
            c[0] = load_px_bytes(irow, x0);
        }
        uint32_t bgw = 0;
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const bool bg = (m[q] & 0xFFu) != 0u && (m[q] & 0xFF00u) != 0u && (m[q] & 0xFF0000u) != 0u;     // bni:88
            if (bg) { c[q] = 0u; bgw |= 1u << (8 * q); }                                                    // bni:91
        }
        uint8_t* orow = a.out.image(im) + (size_t)y * a.out.pitch;
        if (PX == 4) {
            uint32_t cc[4] = {c[0], c[PX > 1 ? 1 : 0], c[PX > 2 ? 2 : 0], c[PX > 3 ? 3 : 0]}, w0, w1, w2;
            pack4(cc, w0, w1, w2);
            uint32_t* op = (uint32_t*)orow + 3 * (size_t)g;
            op[0] = w0; op[1] = w1; op[2] = w2;
            *(uint32_t*)(a.bg + ib + (size_t)y * W + x0) = bgw;
        } else {
            store_px_bytes(orow, x0, c[0]);
            a.bg[ib + (size_t)y * W + x0] = (uint8_t)bgw;
        }
    }
    if (PX == 1) ni_wave_put(lds_list, &lds_cnt, m[0] != 0u, (uint32_t)y * (uint32_t)W + (uint32_t)x0);
    else {                                    // a lane's four pixels are consecutive: slots in pixel order through a wave scan
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < PX; ++q) mine += m[q] != 0u ? 1u : 0u;
        if (__ballot(mine != 0u)) {
            const int lane = threadIdx.x & 63;
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
            uint32_t base = 0;
            if (lane == 63) base = atomicAdd(&lds_cnt, incl);
            base = __shfl(base, 63) + incl - mine;
#pragma unroll
            for (int q = 0; q < PX; ++q) if (m[q] != 0u) lds_list[base++] = (uint32_t)y * (uint32_t)W + (uint32_t)(x0 + q);
        }
    }
    const uint32_t sub = (blockIdx.y * gridDim.x + blockIdx.x) % kNSub;
    ni_emit(a.list_a + ((size_t)im * kNSub + sub) * a.cap_a, a.count + (size_t)(4 * im) * kNSub + sub, lds_list, &lds_cnt, &lds_base);
}

// mark_lower_side (ic:4-49) for one non-black pixel of the mask image.  The samples of kMarchBatch steps are fetched
// together and then looked at in order -- a sample beyond the step that ends the march is simply not used.  Returns the
// marked pixel, or -1.
__device__ __forceinline__ int ni_march_lower_side(const uint8_t* mimg, size_t pitch, uint32_t px, int x, int y, int W, int H, int max_steps = 30)
{
    const float dx0 = ((float)(px & 0xFFu) / 255.0f) * 2.0f - 1.0f;    // ic:10
    const float dy0 = ((float)((px >> 8) & 0xFFu) / 255.0f) * 2.0f - 1.0f;
    const float len = sqrtf(dx0 * dx0 + dy0 * dy0);
    if (!(len > 1e-6f)) return -1;                                     // ic:12
    const float dx = dx0 / len, dy = dy0 / len;
    const float fx = (float)x, fy = (float)y;
    const uint32_t self = __umul24((uint32_t)y, (uint32_t)pitch) + 3u * (uint32_t)x;
    for (int t0 = 1; t0 < max_steps; t0 += kMarchBatch) {
        uint32_t v[kMarchBatch];
        bool in[kMarchBatch];
#pragma unroll
        for (int k = 0; k < kMarchBatch; ++k) {
            uint32_t off;
            in[k] = ni_sample(x, y, dx, dy, (float)(t0 + k), W, H, (uint32_t)pitch, 3u, self, off);
            v[k] = load_px_bytes(mimg + off, 0);
        }
#pragma unroll
        for (int k = 0; k < kMarchBatch; ++k) {
            const int t = t0 + k;
            if (t >= max_steps || !in[k]) return -1;                   // ic:20, 41-42
            if (v[k] != 0u) continue;
            const float bx = rintf(fx + dx * (float)(t - 1)), by = rintf(fy + dy * (float)(t - 1));   // ic:35-39
            return (bx >= 0.0f && by >= 0.0f) ? (int)by * W + (int)bx : -1;
        }
    }
    return -1;
}

// One RGB pixel as an unaligned dword where the byte behind it exists (not for the last pixel of an image), else as bytes.
__device__ __forceinline__ uint32_t ni_load_px(const uint8_t* row, int x, bool dword_ok)
{
    if (dword_ok) { uint32_t v; __builtin_memcpy(&v, row + 3 * (size_t)x, 4); return v & 0xFFFFFFu; }
    return load_px_bytes(row, x);
}

// masked_blur (sr:114-153) of the work image at one pixel: correlation with the 6 x 6 kernel, anchor (3,3), zero border, taps
// row-major in f32; a black pixel stays black.
__device__ __forceinline__ uint32_t ni_masked_blur_px(const uint8_t* work, size_t pitch, int x, int y, int W, int H, const BlurKernel& K)
{
    // all 36 taps are fetched first (a tap outside the image fetches the pixel itself and is not used): one round trip
    uint32_t px[36];
    bool in[36];
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            const int sx = x + kx - 3, sy = y + ky - 3;
            const bool ok = sx >= 0 && sx < W && sy >= 0 && sy < H;
            const int cx = ok ? sx : x, cy = ok ? sy : y;
            in[6 * ky + kx] = ok;
            px[6 * ky + kx] = ni_load_px(work + (size_t)cy * pitch, cx, cy < H - 1 || cx < W - 1);
        }
    }
    if (px[6 * 3 + 3] == 0u) return 0u;
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        if (!in[i]) continue;
        const float k = K.k[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)((px[i] >> (8 * c)) & 0xFFu);
        if (px[i]) wsum = wsum + k;
    }
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = acc[c] / wsum;                          // (the centre is not black: wsum > 0)
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        o |= (uint32_t)v << (8 * c);
    }
    return o;
}

// STAGE 0, a pixel of list A: its requests (bni:104-107 read the 16 taps of a bg pixel's box window) and its lower-side
// march (bni:111).
__device__ __forceinline__ void ni_stage_requests_and_marks(const NiArgs& a, int im, int x, int y)
{
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    const uint8_t* mimg = a.mask.image(im);
    if (a.bg[ib + (size_t)y * W + x] && !(MDVT_DEBUG_SKIP(a) & 2)) {
        uint8_t* need = a.need + ib;
        if (x >= 2 && x + 1 < W && y >= 2 && y + 1 < H) {                 // (no reflection: 4 bytes per row)
#pragma unroll
            for (int dy = -2; dy <= 1; ++dy) {
                uint8_t* r = need + (size_t)(y + dy) * W + (x - 2);
                r[0] = 1; r[1] = 1; r[2] = 1; r[3] = 1;
            }
        } else {
            for (int dy = -2; dy <= 1; ++dy) {
                const size_t ro = (size_t)reflect101(y + dy, H) * W;
                for (int dx = -2; dx <= 1; ++dx) need[ro + reflect101(x + dx, W)] = 1;
            }
        }
    }
    if (MDVT_DEBUG_SKIP(a) & 1) return;
    const int mark = ni_march_lower_side(mimg, a.mask.pitch, load_px_bytes(mimg + (size_t)y * a.mask.pitch, x), x, y, W, H);
    if (mark < 0) return;
    a.marks[ib + (size_t)mark] = 1;
    const int my = mark / W, mx = mark - my * W;
    a.coarse[(size_t)(2 * im) * a.tiles_x * a.tiles_y + (my / kTileH) * a.tiles_x + mx / kTileW] = 1;
}

// STAGE 5, a mark: its diamond of L1 radius 6 = six passes of the 4-neighbour cross (bni:112)
__device__ __forceinline__ void ni_stage_grow(const NiArgs& a, int im, int mx, int my)
{
    const int W = a.W, H = a.H;
    uint8_t* g = a.grown + (size_t)im * H * W;
    for (int ey = -kGrowR; ey <= kGrowR; ++ey) {
        const int yy = my + ey;
        if (yy < 0 || yy >= H) continue;
        const int r = kGrowR - (ey < 0 ? -ey : ey);
        const int x0 = max(mx - r, 0), x1 = min(mx + r, W - 1);
        for (int xx = x0; xx <= x1; ++xx) g[(size_t)yy * W + xx] = 1;
    }
    uint8_t* coarse = a.coarse + (size_t)(2 * im + 1) * a.tiles_x * a.tiles_y;  // the tiles under the diamond's bounding box
    const int tx0 = max(mx - kGrowR, 0) / kTileW, tx1 = min(mx + kGrowR, W - 1) / kTileW;
    const int ty0 = max(my - kGrowR, 0) / kTileH, ty1 = min(my + kGrowR, H - 1) / kTileH;
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) coarse[ty * a.tiles_x + tx] = 1;
}

// STAGE 1, a requested pixel: filled(q) (bni:98-101)
__device__ __forceinline__ void ni_stage_filled(const NiArgs& a, int im, int x, int y)
{
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    const uint8_t* bg = a.bg + ib;
    int sx = x, sy = y;
    bool have = true;
    if (bg[(size_t)y * W + x]) {
        have = false;                                                             // an unfilled hole keeps masked_blur's black
        const uint32_t m = load_px_bytes(a.mask.image(im) + (size_t)y * a.mask.pitch, x);
        const float nx = (((float)(m & 0xFFu) / 255.0f) * 2.0f) - 1.0f;           // bni:94
        const float ny = (((float)((m >> 8) & 0xFFu) / 255.0f) * 2.0f) - 1.0f;
        const float nz = (((float)((m >> 16) & 0xFFu) / 255.0f) * 2.0f) - 1.0f;
        const float len = sqrtf(nx * nx + ny * ny);                               // sr:177
        const bool green = nx == 0.0f && ny == 1.0f && nz == 0.0f;               // sr:182
        if (len > 1e-6f && !green && !(MDVT_DEBUG_SKIP(a) & 4)) {
            const float dx = nx / len, dy = ny / len;
            const float fx = (float)x, fy = (float)y;
            bool done = false;
            for (int t0 = 1; t0 <= 400 && !done; t0 += kHoleBatch) {
                uint8_t hb[kHoleBatch];
                bool in[kHoleBatch];
#pragma unroll
                for (int k = 0; k < kHoleBatch; ++k) {
                    uint32_t off;
                    in[k] = ni_sample(x, y, dx, dy, (float)(t0 + k), W, H, (uint32_t)W, 1u, (uint32_t)y * (uint32_t)W + (uint32_t)x, off);   // sr:205-207
                    hb[k] = bg[off];
                }
#pragma unroll
                for (int k = 0; k < kHoleBatch; ++k) {
                    if (done) break;
                    const int t = t0 + k;
                    if (t > 400 || !in[k]) { done = true; break; }                   // sr:231: the ray left the image
                    if (hb[k]) continue;
                    for (int dt = 2; dt >= 0; --dt) {                                 // sr:220-228
                        const float fo = (float)(t + dt);
                        const float qx = rintf(fx + dx * fo), qy = rintf(fy + dy * fo);
                        if (!ni_in_image(qx, qy, W, H)) continue;
                        if (bg[(size_t)(int)qy * W + (int)qx]) continue;
                        sx = (int)qx; sy = (int)qy; have = true;
                        break;
                    }
                    done = true;
                }
            }
        }
    }
    const uint32_t v = (have && !(MDVT_DEBUG_SKIP(a) & 8)) ? ni_masked_blur_px(a.out.image(im), a.out.pitch, sx, sy, W, H, a.K) : 0u;
    store_px_bytes(a.filled + 3 * (ib + (size_t)y * W), x, v);
}

// STAGE 2, a bg pixel of list A: the 4 x 4 mean of the filled image (bni:104-107)
__device__ __forceinline__ void ni_stage_box(const NiArgs& a, int im, int x, int y)
{
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    if (!a.bg[ib + (size_t)y * W + x]) return;
    uint32_t sum[3] = {0u, 0u, 0u};
#pragma unroll
    for (int dy = -2; dy <= 1; ++dy) {
        const uint8_t* row = a.filled + 3 * (ib + (size_t)reflect101(y + dy, H) * W);
#pragma unroll
        for (int dx = -2; dx <= 1; ++dx) {
            const uint32_t px = ni_load_px(row, reflect101(x + dx, W), true);      // (the buffer is padded)
            sum[0] += px & 0xFFu; sum[1] += (px >> 8) & 0xFFu; sum[2] += (px >> 16) & 0xFFu;
        }
    }
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t q = sum[c] >> 4;                                // cvRound(sum / 16): half to even
        const uint32_t rem = sum[c] & 15u;
        if (rem > 8u || (rem == 8u && (q & 1u))) ++q;
        o |= q << (8 * c);
    }
    store_px_bytes(a.out.image(im) + (size_t)y * a.out.pitch, x, o);
}

// STAGE 3, a grown pixel: blur_under_mask (bni:118, 46-85) into the side buffer
__device__ __forceinline__ void ni_stage_blur_under(const NiArgs& a, int im, int x, int y)
{
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    const uint8_t* work = a.out.image(im);
    const uint8_t* grown = a.grown + ib;
    // flags and pixels of all 36 taps are fetched first (a tap outside the image fetches the pixel itself and is not used)
    uint32_t px[36];
    bool use[36];
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            const int sx = x + kx - 3, sy = y + ky - 3;
            const bool ok = sx >= 0 && sx < W && sy >= 0 && sy < H;
            const int cx = ok ? sx : x, cy = ok ? sy : y;
            use[6 * ky + kx] = ok && grown[(size_t)cy * W + cx] != 0;               // bni:68: img_f * m
            px[6 * ky + kx] = ni_load_px(work + (size_t)cy * a.out.pitch, cx, cy < H - 1 || cx < W - 1);
        }
    }
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        if (!use[i]) continue;
        const float k = a.K.k[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)((px[i] >> (8 * c)) & 0xFFu);
        wsum = wsum + k;
    }
    uint32_t o = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = acc[c] / wsum;                                     // (the pixel itself is grown: wsum > 0)
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        o |= (uint32_t)v << (8 * c);
    }
    store_px_bytes(a.filled + 3 * (ib + (size_t)y * W), x, o);
}

// one listed pixel per lane: blockIdx.y = sub-list, blockIdx.z = image
template <int STAGE>
__global__ void __launch_bounds__(256) k_ni_run(NiArgs a)
{
    const int im = blockIdx.z, sub = blockIdx.y, W = a.W;
    const size_t ib = (size_t)im * a.H * W;
    constexpr int L = (STAGE == 0 || STAGE == 2) ? 0 : STAGE == 1 ? 1 : STAGE == 5 ? 3 : 2;
    const uint32_t* list = L == 0 ? a.list_a + ((size_t)im * kNSub + sub) * a.cap_a : a.list_b + ((size_t)im * kNSub + sub) * a.cap_b;
    const uint32_t n = a.count[(size_t)(4 * im + L) * kNSub + sub];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t idx = list[i];
        const int y = (int)(idx / (uint32_t)W), x = (int)(idx - (uint32_t)y * (uint32_t)W);
        if (STAGE == 0) ni_stage_requests_and_marks(a, im, x, y);
        else if (STAGE == 1) ni_stage_filled(a, im, x, y);
        else if (STAGE == 2) ni_stage_box(a, im, x, y);
        else if (STAGE == 3) ni_stage_blur_under(a, im, x, y);
        else if (STAGE == 5) ni_stage_grow(a, im, x, y);
        else store_px_bytes(a.out.image(im) + (size_t)y * a.out.pitch, x, load_px_bytes(a.filled + 3 * ib, (int)idx));
    }
}

// the set pixels of a byte plane -> a list.  WHICH 0: need -> list B; 1: grown -> list C; 2: marks -> list M (1 and 2: only the
// tiles their coarse map names)
template <int WHICH>
__global__ void __launch_bounds__(256) k_ni_collect(NiArgs a)
{
    __shared__ uint32_t lds_list[kTileW * kTileH];
    __shared__ uint32_t lds_cnt, lds_base;
    const int W = a.W, H = a.H, im = blockIdx.z;
    const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (WHICH != 0 && !a.coarse[(size_t)(2 * im + (WHICH == 1 ? 1 : 0)) * a.tiles_y * a.tiles_x + tile]) return;     // (workgroup-uniform)
    const size_t ib = (size_t)im * H * W;
    const uint8_t* plane = (WHICH == 0 ? a.need : WHICH == 1 ? a.grown : a.marks) + ib;
    if (threadIdx.x == 0) lds_cnt = 0u;
    __syncthreads();
    // a wave looks at 64 consecutive pixels of a tile row at a time (rows wave, wave + 4, ...: 8 rows per thread)
    const int x = blockIdx.x * kTileW + (threadIdx.x & 127);
#pragma unroll
    for (int k = 0; k < kTileH / 2; ++k) {
        const int y = blockIdx.y * kTileH + (threadIdx.x >> 7) + 2 * k;
        const bool f = y < H && x < W && plane[(size_t)y * W + x] != 0;
        ni_wave_put(lds_list, &lds_cnt, f, (uint32_t)y * (uint32_t)W + (uint32_t)x);
    }
    const uint32_t sub = tile % kNSub;
    ni_emit(a.list_b + ((size_t)im * kNSub + sub) * a.cap_b, a.count + (size_t)(4 * im + 1 + WHICH) * kNSub + sub, lds_list, &lds_cnt, &lds_base);
}

// ---- sr:809-812 (--do_basic_infill) for whole batches: the holes of the stereo frames filled along the finished infill mask ----
// The reference calls infill_using_normals(image, bg_mask, mask * 2 - 1) per eye; its sources are never hole pixels, so the
// image can be filled in place, and the normal of a hole pixel is needed by that pixel alone: ((u8 / 255) * 2) - 1 on the spot
// instead of a float image.  Same scheme as above: a dense pass lists the hole pixels, then one lane per listed pixel.
struct HfArgs {
    ImageSet img, hole, mask;         // img: filled in place; hole: u8 plane, non-zero = hole; mask: the finished infill-mask image
    uint32_t* count;                  // [n][kNSub]
    uint32_t* list;                   // [n][kNSub][cap]
    uint32_t cap;
    int W, H, max_steps;
};

__global__ void __launch_bounds__(256) k_hf_collect(HfArgs a)
{
    __shared__ uint32_t lds_list[kTileW * kTileH];
    __shared__ uint32_t lds_cnt, lds_base;
    const int W = a.W, H = a.H, im = blockIdx.z;
    const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
    const uint8_t* plane = a.hole.image(im);
    if (threadIdx.x == 0) lds_cnt = 0u;
    __syncthreads();
    const int x = blockIdx.x * kTileW + (threadIdx.x & 127);
#pragma unroll
    for (int k = 0; k < kTileH / 2; ++k) {
        const int y = blockIdx.y * kTileH + (threadIdx.x >> 7) + 2 * k;
        const bool f = y < H && x < W && plane[(size_t)y * a.hole.pitch + x] != 0;
        ni_wave_put(lds_list, &lds_cnt, f, (uint32_t)y * (uint32_t)W + (uint32_t)x);
    }
    const uint32_t sub = tile % kNSub;
    ni_emit(a.list + ((size_t)im * kNSub + sub) * a.cap, a.count + (size_t)im * kNSub + sub, lds_list, &lds_cnt, &lds_base);
}

__global__ void __launch_bounds__(256) k_hf_run(HfArgs a)
{
    const int im = blockIdx.z, sub = blockIdx.y, W = a.W, H = a.H;
    const uint32_t* list = a.list + ((size_t)im * kNSub + sub) * a.cap;
    const uint32_t n = a.count[(size_t)im * kNSub + sub];
    const uint8_t* hole = a.hole.image(im);
    const size_t hp = a.hole.pitch;
    uint8_t* img = a.img.image(im);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t idx = list[i];
        const int y = (int)(idx / (uint32_t)W), x = (int)(idx - (uint32_t)y * (uint32_t)W);
        const uint32_t m = load_px_bytes(a.mask.image(im) + (size_t)y * a.mask.pitch, x);
        const float nx = (((float)(m & 0xFFu) / 255.0f) * 2.0f) - 1.0f;           // sr:808 (.../255.0) and sr:810 (*2 - 1), f32
        const float ny = (((float)((m >> 8) & 0xFFu) / 255.0f) * 2.0f) - 1.0f;
        const float nz = (((float)((m >> 16) & 0xFFu) / 255.0f) * 2.0f) - 1.0f;
        const float len = sqrtf(nx * nx + ny * ny);                               // sr:177
        const bool green = nx == 0.0f && ny == 1.0f && nz == 0.0f;               // sr:182
        if (!(len > 1e-6f) || green) continue;
        const float dx = nx / len, dy = ny / len;
        const float fx = (float)x, fy = (float)y;
        bool done = false;
        for (int t0 = 1; t0 <= a.max_steps && !done; t0 += kHoleBatch) {
            uint8_t hb[kHoleBatch];
            bool in[kHoleBatch];
#pragma unroll
            for (int k = 0; k < kHoleBatch; ++k) {
                uint32_t off;
                in[k] = ni_sample(x, y, dx, dy, (float)(t0 + k), W, H, (uint32_t)hp, 1u, (uint32_t)y * (uint32_t)hp + (uint32_t)x, off);   // sr:205-207
                hb[k] = hole[off];
            }
#pragma unroll
            for (int k = 0; k < kHoleBatch; ++k) {
                if (done) break;
                const int t = t0 + k;
                if (t > a.max_steps || !in[k]) { done = true; break; }               // sr:231: the ray left the image
                if (hb[k]) continue;
                for (int dt = 2; dt >= 0; --dt) {                                 // sr:220-228
                    const float fo = (float)(t + dt);
                    const float qx = rintf(fx + dx * fo), qy = rintf(fy + dy * fo);
                    if (!ni_in_image(qx, qy, W, H)) continue;
                    if (hole[(size_t)(int)qy * hp + (int)qx]) continue;
                    store_px_bytes(img + (size_t)y * a.img.pitch, x, load_px_bytes(img + (size_t)(int)qy * a.img.pitch, (int)qx));
                    break;
                }
                done = true;
            }
        }
    }
}

// ---- the two stand-alone entry points on the same scheme: mdvt_infill_using_normals (float normals, sr:155-240) and
//      mdvt_mark_lower_side (ic:4-49), one image per call ----
struct SoloArgs {
    const uint8_t* src; size_t src_pitch;        // infill: the colour image; lower side: the normal-coloured mask image
    const uint8_t* hole; size_t hole_pitch;      // infill only
    const float* normal; size_t normal_pitch;    // infill only (bytes)
    uint8_t* out; size_t out_pitch;
    uint32_t* count;                             // [kNSub]
    uint32_t* list;                              // [kNSub][cap]
    uint32_t cap;
    int W, H, max_steps;
};

// WHICH 0: the hole pixels; 1: the non-black pixels of the mask image
template <int WHICH>
__global__ void __launch_bounds__(256) k_solo_collect(SoloArgs a)
{
    __shared__ uint32_t lds_list[kTileW * kTileH];
    __shared__ uint32_t lds_cnt, lds_base;
    const int W = a.W, H = a.H;
    const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0) lds_cnt = 0u;
    __syncthreads();
    const int x = blockIdx.x * kTileW + (threadIdx.x & 127);
#pragma unroll
    for (int k = 0; k < kTileH / 2; ++k) {
        const int y = blockIdx.y * kTileH + (threadIdx.x >> 7) + 2 * k;
        bool f = false;
        if (y < H && x < W) f = WHICH == 0 ? a.hole[(size_t)y * a.hole_pitch + x] != 0 : load_px_bytes(a.src + (size_t)y * a.src_pitch, x) != 0u;
        ni_wave_put(lds_list, &lds_cnt, f, (uint32_t)y * (uint32_t)W + (uint32_t)x);
    }
    const uint32_t sub = tile % kNSub;
    ni_emit(a.list + (size_t)sub * a.cap, a.count + sub, lds_list, &lds_cnt, &lds_base);
}

template <int WHICH>
__global__ void __launch_bounds__(256) k_solo_run(SoloArgs a)
{
    const int sub = blockIdx.y, W = a.W, H = a.H;
    const uint32_t* list = a.list + (size_t)sub * a.cap;
    const uint32_t n = a.count[sub];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t idx = list[i];
        const int y = (int)(idx / (uint32_t)W), x = (int)(idx - (uint32_t)y * (uint32_t)W);
        if (WHICH == 1) {
            const int mark = ni_march_lower_side(a.src, a.src_pitch, load_px_bytes(a.src + (size_t)y * a.src_pitch, x), x, y, W, H, a.max_steps);
            if (mark >= 0) store_px_bytes(a.out + (size_t)(mark / W) * a.out_pitch, mark % W, 0xFF0000u);      // (0, 0, 255), ic:46
            continue;
        }
        const float* nrm = (const float*)((const uint8_t*)a.normal + (size_t)y * a.normal_pitch) + 3 * (size_t)x;
        const float nx = nrm[0], ny = nrm[1], nz = nrm[2];
        const float len = sqrtf(nx * nx + ny * ny);                               // sr:177
        const bool green = nx == 0.0f && ny == 1.0f && nz == 0.0f;               // sr:182
        if (!(len > 1e-6f) || green) continue;
        const float dx = nx / len, dy = ny / len;
        const float fx = (float)x, fy = (float)y;
        const uint8_t* hole = a.hole;
        const size_t hp = a.hole_pitch;
        bool done = false;
        for (int t0 = 1; t0 <= a.max_steps && !done; t0 += kHoleBatch) {
            uint8_t hb[kHoleBatch];
            bool in[kHoleBatch];
#pragma unroll
            for (int k = 0; k < kHoleBatch; ++k) {
                uint32_t off;
                in[k] = ni_sample(x, y, dx, dy, (float)(t0 + k), W, H, (uint32_t)hp, 1u, (uint32_t)y * (uint32_t)hp + (uint32_t)x, off);   // sr:205-207
                hb[k] = hole[off];
            }
#pragma unroll
            for (int k = 0; k < kHoleBatch; ++k) {
                if (done) break;
                const int t = t0 + k;
                if (t > a.max_steps || !in[k]) { done = true; break; }               // sr:231: the ray left the image
                if (hb[k]) continue;
                for (int dt = 2; dt >= 0; --dt) {                                 // sr:220-228
                    const float fo = (float)(t + dt);
                    const float qx = rintf(fx + dx * fo), qy = rintf(fy + dy * fo);
                    if (!ni_in_image(qx, qy, W, H)) continue;
                    if (hole[(size_t)(int)qy * hp + (int)qx]) continue;
                    store_px_bytes(a.out + (size_t)y * a.out_pitch, x, load_px_bytes(a.src + (size_t)(int)qy * a.src_pitch, (int)qx));
                    break;
                }
                done = true;
            }
        }
    }
}

struct NiLayout { uint32_t cap_a, cap_b; int tiles_x, tiles_y; size_t ncoarse, bytes_lists, bytes_zero, bytes_total; };

// cap_a covers both shapes of k_ni_prep (128 threads x 4 px or x 1 px per workgroup)
NiLayout ni_layout(int n, int W, int H)
{
    NiLayout l;
    const size_t npx = (size_t)W * H;
    auto cap = [](size_t producers, size_t px_each) { return (uint32_t)(((producers + kNSub - 1) / kNSub) * px_each); };
    const uint32_t c4 = cap((size_t)((W / 4 + 127) / 128 + 1) * H, 512), c1 = cap((size_t)((W + 127) / 128) * H, 128);
    l.cap_a = c4 > c1 ? c4 : c1;
    l.tiles_x = (W + kTileW - 1) / kTileW; l.tiles_y = (H + kTileH - 1) / kTileH;
    l.cap_b = cap((size_t)l.tiles_x * l.tiles_y, (size_t)kTileW * kTileH);
    l.bytes_lists = (size_t)n * kNSub * ((size_t)l.cap_a + l.cap_b) * sizeof(uint32_t);
    l.ncoarse = ((size_t)n * 2 * l.tiles_x * l.tiles_y + 3) & ~(size_t)3;
    l.bytes_zero = (size_t)n * 4 * kNSub * sizeof(uint32_t) + l.ncoarse + 3 * (size_t)n * npx;
    l.bytes_total = l.bytes_lists + l.bytes_zero + (size_t)n * npx * 4 + 64;
    return l;
}

}  // namespace

// sub-lists (~4 + 4 B/px), filled 3, bg / need / marks / grown 1 each + counters, coarse maps
size_t normal_infill_workspace_bytes(int n, int W, int H) { return ni_layout(n, W, H).bytes_total; }

// workspace: the same allocation as launch_normal_infill's (its list B and counters)
hipError_t launch_infill_mask_normals(const ImageSet& img, const ImageSet& hole, const ImageSet& mask, uint8_t* workspace, int n, int W, int H,
                                      int max_steps, hipStream_t s)
{
    const size_t npx = (size_t)W * H;
    const NiLayout l = ni_layout(n, W, H);
    HfArgs a;
    a.img = img; a.hole = hole; a.mask = mask;
    a.W = W; a.H = H; a.max_steps = max_steps;
    a.cap = l.cap_b;
    a.list = (uint32_t*)workspace + (size_t)n * kNSub * l.cap_a;
    a.count = a.list + (size_t)n * kNSub * l.cap_b;
    hipError_t e = hipMemsetAsync(a.count, 0, (size_t)n * kNSub * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    unsigned per_sub = (unsigned)((npx / kNSub / 16 + 255) / 256);
    per_sub = per_sub < 1 ? 1 : (per_sub > 16 ? 16 : per_sub);
    hipLaunchKernelGGL(k_hf_collect, dim3(l.tiles_x, l.tiles_y, n), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_hf_run, dim3(per_sub, kNSub, n), dim3(256), 0, s, a);
    return hipGetLastError();
}

static hipError_t solo_launch(int which, SoloArgs a, uint8_t* workspace, hipStream_t s)
{
    const NiLayout l = ni_layout(1, a.W, a.H);
    a.cap = l.cap_b;
    a.list = (uint32_t*)workspace + (size_t)kNSub * l.cap_a;
    a.count = a.list + (size_t)kNSub * l.cap_b;
    hipError_t e = hipMemsetAsync(a.count, 0, (size_t)kNSub * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    unsigned per_sub = (unsigned)(((size_t)a.W * a.H / kNSub / 16 + 255) / 256);
    per_sub = per_sub < 1 ? 1 : (per_sub > 16 ? 16 : per_sub);
    const dim3 tiles(l.tiles_x, l.tiles_y), lanes(per_sub, kNSub);
    if (which == 0) {
        hipLaunchKernelGGL(k_solo_collect<0>, tiles, dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_solo_run<0>, lanes, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(k_solo_collect<1>, tiles, dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_solo_run<1>, lanes, dim3(256), 0, s, a);
    }
    return hipGetLastError();
}

// mdvt_infill_using_normals: out = the colour image with its holes filled (workspace: normal_infill_workspace_bytes(1, W, H))
hipError_t launch_infill_normals(const uint8_t* color, size_t color_pitch, const uint8_t* hole, size_t hole_pitch,
                                 const float* normal, size_t normal_pitch, uint8_t* out, size_t out_pitch, int W, int H,
                                 int max_steps, uint8_t* workspace, hipStream_t s)
{
    hipError_t e = hipMemcpy2DAsync(out, out_pitch, color, color_pitch, (size_t)3 * W, (size_t)H, hipMemcpyDeviceToDevice, s);     // sr:172
    if (e != hipSuccess) return e;
    SoloArgs a{};
    a.src = color; a.src_pitch = color_pitch; a.hole = hole; a.hole_pitch = hole_pitch; a.normal = normal; a.normal_pitch = normal_pitch;
    a.out = out; a.out_pitch = out_pitch; a.W = W; a.H = H; a.max_steps = max_steps;
    return solo_launch(0, a, workspace, s);
}

// mdvt_mark_lower_side: out = (0,0,255) at the marks, black elsewhere
hipError_t launch_mark_lower_side(const uint8_t* img, size_t img_pitch, uint8_t* out, size_t out_pitch, int W, int H,
                                  int max_steps, uint8_t* workspace, hipStream_t s)
{
    hipError_t e = hipMemset2DAsync(out, out_pitch, 0, (size_t)3 * W, (size_t)H, s);
    if (e != hipSuccess) return e;
    SoloArgs a{};
    a.src = img; a.src_pitch = img_pitch; a.out = out; a.out_pitch = out_pitch; a.W = W; a.H = H; a.max_steps = max_steps;
    return solo_launch(1, a, workspace, s);
}

hipError_t launch_normal_infill(const ImageSet& img, const ImageSet& mask, const ImageSet& out, uint8_t* workspace, int n, int W, int H,
                                const BlurKernel& K, hipStream_t s)
{
    const size_t npx = (size_t)W * H;
    const NiLayout l = ni_layout(n, W, H);
    NiArgs a;
    a.img = img; a.mask = mask; a.out = out;
    a.W = W; a.H = H; a.K = K;
    a.tiles_x = l.tiles_x; a.tiles_y = l.tiles_y;
    a.cap_a = l.cap_a; a.cap_b = l.cap_b;
    // (the lists first: 4-byte aligned whatever W and H are; then everything that starts out as zero, in one block)
    a.list_a = (uint32_t*)workspace;
    a.list_b = a.list_a + (size_t)n * kNSub * l.cap_a;
    a.count = a.list_b + (size_t)n * kNSub * l.cap_b;
    a.coarse = (uint8_t*)(a.count + (size_t)n * 4 * kNSub);
    a.need = a.coarse + l.ncoarse;
    a.marks = a.need + (size_t)n * npx;
    a.grown = a.marks + (size_t)n * npx;
    uint8_t* zero_end = a.grown + (size_t)n * npx;
    a.bg = zero_end;
    a.filled = a.bg + (size_t)n * npx;
    a.debug_skip = 0;
    if (const char* ev = tuning_env(TUNE_NI_SKIP)) a.debug_skip = atoi(ev);
    hipError_t e = hipMemsetAsync(a.count, 0, (size_t)(zero_end - (uint8_t*)a.count), s);
    if (e != hipSuccess) return e;
    auto dwords = [](const ImageSet& i) { return (((uintptr_t)i.base | i.pitch | i.stride) & 3) == 0; };
    const bool vec = W % 4 == 0 && dwords(img) && dwords(mask) && dwords(out);
    if (vec) hipLaunchKernelGGL(k_ni_prep<4>, dim3((W / 4 + 127) / 128, H, n), dim3(128), 0, s, a);
    else hipLaunchKernelGGL(k_ni_prep<1>, dim3((W + 127) / 128, H, n), dim3(128), 0, s, a);
    // lanes per sub-list: a sixteenth of what it could hold at most (holes are a few per cent of a frame; the rest loops)
    unsigned per_sub = (unsigned)((npx / kNSub / 16 + 255) / 256);
    per_sub = per_sub < 1 ? 1 : (per_sub > 16 ? 16 : per_sub);
    const dim3 tiles(a.tiles_x, a.tiles_y, n), lanes(per_sub, kNSub, n);
    hipLaunchKernelGGL(k_ni_run<0>, lanes, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_collect<2>, tiles, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_run<5>, lanes, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_collect<0>, tiles, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_run<1>, lanes, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_run<2>, lanes, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_collect<1>, tiles, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_run<3>, lanes, dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_ni_run<4>, lanes, dim3(256), 0, s, a);
    if (tuning_env(TUNE_NI_DUMP)) {            // tuning hook: list sizes of this call on stderr
        std::vector<uint32_t> c((size_t)n * 4 * kNSub);
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
        if ((e = hipMemcpy(c.data(), a.count, c.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess) return e;
        for (int im = 0; im < n; ++im) {
            uint32_t tot[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
            for (int l = 0; l < 4; ++l)
                for (int k = 0; k < kNSub; ++k) { const uint32_t v = c[((size_t)4 * im + l) * kNSub + k]; tot[l] += v; mx[l] = v > mx[l] ? v : mx[l]; }
            fprintf(stderr, "image %d: list A %u (largest sub-list %u), B %u (%u), C %u (%u), M %u (%u); lanes per sub-list %u\n", im, tot[0], mx[0],
                    tot[1], mx[1], tot[2], mx[2], tot[3], mx[3], per_sub * 256);
        }
    }
    return hipGetLastError();
}

}  // namespace mdvt
