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
#include "mdvt_device.h"

namespace mdvt {

namespace {

struct NiArgs {
    ImageSet img, mask, out;          // caller's images (u8 RGB rows)
    uint8_t* work;                    // [n][H*W*3]
    uint8_t* filled;                  // [n][H*W*3] valid where need != 0
    uint8_t* bg;                      // [n][H*W]
    uint8_t* need;                    // [n][H*W]
    uint8_t* grown;                   // [n][H*W]
    int W, H;
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

// bni:88-91 and the reset of the two scatter planes
__global__ void __launch_bounds__(256) k_ni_prep(NiArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= a.W) return;
    const size_t o = ((size_t)im * a.H + y) * a.W + x;
    const uint32_t m = load_px_bytes(a.mask.image(im) + (size_t)y * a.mask.pitch, x);
    const bool bg = (m & 0xFFu) != 0u && (m & 0xFF00u) != 0u && (m & 0xFF0000u) != 0u;
    const uint32_t c = bg ? 0u : load_px_bytes(a.img.image(im) + (size_t)y * a.img.pitch, x);
    a.bg[o] = bg ? 1 : 0;
    a.need[o] = 0;
    a.grown[o] = 0;
    store_px_bytes(a.work + 3 * (o - x), x, c);
}

// every bg pixel asks for the 16 taps of its box window (bni:104 read at bni:107)
__global__ void __launch_bounds__(256) k_ni_request(NiArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= a.W) return;
    const size_t ib = (size_t)im * a.H * a.W;
    if (!a.bg[ib + (size_t)y * a.W + x]) return;
#pragma unroll
    for (int dy = -2; dy <= 1; ++dy) {
        const size_t ro = ib + (size_t)reflect101(y + dy, a.H) * a.W;
#pragma unroll
        for (int dx = -2; dx <= 1; ++dx) a.need[ro + reflect101(x + dx, a.W)] = 1;
    }
}

// masked_blur (sr:114-153) of the work image at one pixel: correlation with the 6 x 6 kernel, anchor (3,3), zero border, taps
// row-major in f32; a black pixel stays black.
__device__ __forceinline__ uint32_t ni_masked_blur_px(const uint8_t* work, int x, int y, int W, int H, const BlurKernel& K)
{
    const uint32_t centre = load_px_bytes(work + (size_t)y * 3 * W, x);
    if (centre == 0u) return 0u;
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        const int sy = y + ky - 3;
        if (sy < 0 || sy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            const int sx = x + kx - 3;
            if (sx < 0 || sx >= W) continue;
            const uint32_t px = load_px_bytes(work + (size_t)sy * 3 * W, sx);
            const float k = K.k[6 * ky + kx];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)((px >> (8 * c)) & 0xFFu);
            if (px) wsum = wsum + k;
        }
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

// filled(q) for the requested pixels (bni:98-101)
__global__ void __launch_bounds__(256) k_ni_filled(NiArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= a.W) return;
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    const size_t o = ib + (size_t)y * W + x;
    if (!a.need[o]) return;
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
        if (len > 1e-6f && !green) {
            const float dx = nx / len, dy = ny / len;
            const float fx = (float)x, fy = (float)y;
            for (int t = 1; t <= 400; ++t) {
                const float ft = (float)t;
                const float rx = rintf(fx + dx * ft), ry = rintf(fy + dy * ft);  // sr:205-207
                if (!ni_in_image(rx, ry, W, H)) break;                           // sr:231
                if (bg[(size_t)(int)ry * W + (int)rx]) continue;
                for (int dt = 2; dt >= 0; --dt) {                                 // sr:220-228
                    const float fo = (float)(t + dt);
                    const float qx = rintf(fx + dx * fo), qy = rintf(fy + dy * fo);
                    if (!ni_in_image(qx, qy, W, H)) continue;
                    if (bg[(size_t)(int)qy * W + (int)qx]) continue;
                    sx = (int)qx; sy = (int)qy; have = true;
                    break;
                }
                break;
            }
        }
    }
    const uint32_t v = have ? ni_masked_blur_px(a.work + 3 * ib, sx, sy, W, H, a.K) : 0u;
    store_px_bytes(a.filled + 3 * (ib + (size_t)y * W), x, v);
}

// bni:104-107: the bg pixels take the 4 x 4 mean of the filled image
__global__ void __launch_bounds__(256) k_ni_box(NiArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= a.W) return;
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    if (!a.bg[ib + (size_t)y * W + x]) return;
    uint32_t sum[3] = {0u, 0u, 0u};
#pragma unroll
    for (int dy = -2; dy <= 1; ++dy) {
        const uint8_t* row = a.filled + 3 * (ib + (size_t)reflect101(y + dy, H) * W);
#pragma unroll
        for (int dx = -2; dx <= 1; ++dx) {
            const uint32_t px = load_px_bytes(row, reflect101(x + dx, W));
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
    store_px_bytes(a.work + 3 * (ib + (size_t)y * W), x, o);
}

// bni:111-115: mark_lower_side (ic:4-49) on the mask image; a mark sets its diamond of L1 radius 6 straight away
__global__ void __launch_bounds__(256) k_ni_marks(NiArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= a.W) return;
    const int W = a.W, H = a.H;
    const uint8_t* mimg = a.mask.image(im);
    const uint32_t px = load_px_bytes(mimg + (size_t)y * a.mask.pitch, x);
    if (px == 0u) return;                                              // ic:7
    const float dx0 = ((float)(px & 0xFFu) / 255.0f) * 2.0f - 1.0f;    // ic:10
    const float dy0 = ((float)((px >> 8) & 0xFFu) / 255.0f) * 2.0f - 1.0f;
    const float len = sqrtf(dx0 * dx0 + dy0 * dy0);
    if (!(len > 1e-6f)) return;                                        // ic:12
    const float dx = dx0 / len, dy = dy0 / len;
    const float fx = (float)x, fy = (float)y;
    for (int t = 1; t < 30; ++t) {
        const float rx = rintf(fx + dx * (float)t), ry = rintf(fy + dy * (float)t);
        if (!ni_in_image(rx, ry, W, H)) return;                        // ic:41-42
        if (load_px_bytes(mimg + (size_t)(int)ry * a.mask.pitch, (int)rx) != 0u) continue;
        const float bx = rintf(fx + dx * (float)(t - 1)), by = rintf(fy + dy * (float)(t - 1));   // ic:35-39
        if (!(bx >= 0.0f && by >= 0.0f)) return;
        const int mx = (int)bx, my = (int)by;
        uint8_t* g = a.grown + (size_t)im * H * W;
        for (int ey = -6; ey <= 6; ++ey) {
            const int yy = my + ey;
            if (yy < 0 || yy >= H) continue;
            const int r = 6 - (ey < 0 ? -ey : ey);
            const int x0 = max(mx - r, 0), x1 = min(mx + r, W - 1);
            for (int xx = x0; xx <= x1; ++xx) g[(size_t)yy * W + xx] = 1;
        }
        return;
    }
}

// bni:118: blur_under_mask inside the grown plane, the image itself elsewhere
__global__ void __launch_bounds__(256) k_ni_final(NiArgs a)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, im = blockIdx.z;
    if (x >= a.W) return;
    const int W = a.W, H = a.H;
    const size_t ib = (size_t)im * H * W;
    const uint8_t* work = a.work + 3 * ib;
    const uint8_t* grown = a.grown + ib;
    uint32_t o = load_px_bytes(work + (size_t)y * 3 * W, x);
    if (grown[(size_t)y * W + x]) {
        float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 6; ++ky) {
            const int sy = y + ky - 3;
            if (sy < 0 || sy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 6; ++kx) {
                const int sx = x + kx - 3;
                if (sx < 0 || sx >= W) continue;
                if (!grown[(size_t)sy * W + sx]) continue;               // bni:68: img_f * m
                const uint32_t px = load_px_bytes(work + (size_t)sy * 3 * W, sx);
                const float k = a.K.k[6 * ky + kx];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)((px >> (8 * c)) & 0xFFu);
                wsum = wsum + k;
            }
        }
        o = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc[c] / wsum;
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            o |= (uint32_t)v << (8 * c);
        }
    }
    store_px_bytes(a.out.image(im) + (size_t)y * a.out.pitch, x, o);
}

}  // namespace

size_t normal_infill_workspace_bytes(int n, int W, int H) { return (size_t)n * W * H * 9; }

hipError_t launch_normal_infill(const ImageSet& img, const ImageSet& mask, const ImageSet& out, uint8_t* workspace, int n, int W, int H,
                                const BlurKernel& K, hipStream_t s)
{
    const size_t npx = (size_t)W * H;
    NiArgs a;
    a.img = img; a.mask = mask; a.out = out;
    a.work = workspace;
    a.filled = a.work + (size_t)n * npx * 3;
    a.bg = a.filled + (size_t)n * npx * 3;
    a.need = a.bg + (size_t)n * npx;
    a.grown = a.need + (size_t)n * npx;
    a.W = W; a.H = H; a.K = K;
    const dim3 grid((W + 255) / 256, H, n), block(256);
    hipLaunchKernelGGL(k_ni_prep, grid, block, 0, s, a);
    hipLaunchKernelGGL(k_ni_request, grid, block, 0, s, a);
    hipLaunchKernelGGL(k_ni_filled, grid, block, 0, s, a);
    hipLaunchKernelGGL(k_ni_box, grid, block, 0, s, a);
    hipLaunchKernelGGL(k_ni_marks, grid, block, 0, s, a);
    hipLaunchKernelGGL(k_ni_final, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace mdvt
