// mdvt_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the stereo-rerender hot path.
//
// Replaces the NumPy + Open3D/OpenGL stages of the reference's frame loop (stereo_rerender.py:512-907):
//   decode (dfh:63-75, 13-24) -> master scale (sr:541) -> unproject (dmt:1112-1133) -> eye/pose
//   transform (sr:615-619, 724-725, 832-836) -> z-buffered render (dmt:1422-1572) -> colour-key hole
//   mask (sr:740, 793) -> edge-point splat (sr:745-814).
//
// The path is an HBM-bound gather/scatter: no MFMA.  What matters is (i) every HBM byte is read and
// written once, coalesced (12 B/lane dwordx3 = 768 contiguous bytes per wave), (ii) the z-buffer
// lives in LDS for the row-local (pure stereo shift) case so the atomics never leave the CU,
// (iii) one launch covers a whole batch of frames (n_frames * H workgroups >> 256 CUs).
//
// Compiled with -ffp-contract=off: see the arithmetic decree in mdvt_device.h / DESIGN.md.
#include "mdvt_device.h"

namespace mdvt {

// =================================================================================================
// depth codec (dfh)
// =================================================================================================

__global__ void k_decode_depth(const uint8_t* __restrict__ rgb, size_t rgb_pitch, float* __restrict__ out,
                               size_t out_pitch, int W, int H, float mult, float scale)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= W || i >= H) return;
    const uint32_t px = load_px_bytes(rgb + (size_t)i * rgb_pitch, j);
    float* orow = (float*)((uint8_t*)out + (size_t)i * out_pitch);
    orow[j] = decode_z(code16_of(px), mult, scale);
}

// 4 pixels per thread: 12 B coalesced load, 16 B coalesced store.
__global__ void k_decode_depth4(const uint8_t* __restrict__ rgb, size_t rgb_pitch, float* __restrict__ out,
                                size_t out_pitch, int W4, int H, float mult, float scale)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (g >= W4 || i >= H) return;
    const uint32_t* src = (const uint32_t*)(rgb + (size_t)i * rgb_pitch) + 3 * (size_t)g;
    uint32_t px[4];
    unpack4(src[0], src[1], src[2], px);
    float4 z;
    z.x = decode_z(code16_of(px[0]), mult, scale);
    z.y = decode_z(code16_of(px[1]), mult, scale);
    z.z = decode_z(code16_of(px[2]), mult, scale);
    z.w = decode_z(code16_of(px[3]), mult, scale);
    ((float4*)((uint8_t*)out + (size_t)i * out_pitch))[g] = z;
}

hipError_t launch_decode_depth(const uint8_t* rgb, size_t rgb_pitch, float* out, size_t out_pitch, int W, int H,
                               float mult, float scale, hipStream_t s)
{
    const bool vec = (W % 4 == 0) && (rgb_pitch % 4 == 0) && (out_pitch % 16 == 0) &&
                     ((uintptr_t)rgb % 4 == 0) && ((uintptr_t)out % 16 == 0);
    if (vec) {
        const int W4 = W / 4;
        dim3 grid((W4 + 255) / 256, H);
        hipLaunchKernelGGL(k_decode_depth4, grid, dim3(256), 0, s, rgb, rgb_pitch, out, out_pitch, W4, H, mult, scale);
    } else {
        dim3 grid((W + 255) / 256, H);
        hipLaunchKernelGGL(k_decode_depth, grid, dim3(256), 0, s, rgb, rgb_pitch, out, out_pitch, W, H, mult, scale);
    }
    return hipGetLastError();
}

// dfh:5-11: clip to [0,max] in f32, f64 multiply by 255^4/max, truncate to u32; dfh:53-55: R = G = byte 3,
// B = byte 2.
__global__ void k_encode_depth(const float* __restrict__ depth, size_t depth_pitch, uint8_t* __restrict__ rgb,
                               size_t rgb_pitch, int W, int H, double multi, float fmax_depth, int bgr)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= W || i >= H) return;
    float d = ((const float*)((const uint8_t*)depth + (size_t)i * depth_pitch))[j];
    if (d > fmax_depth) d = fmax_depth;
    if (d < 0.0f) d = 0.0f;
    const double e = multi * (double)d;
    const uint32_t code = (e >= 0.0 && e < 4294967296.0) ? (uint32_t)e : 0u;     // NaN -> 0
    const uint32_t hi = code >> 24, lo = (code >> 16) & 0xFFu;
    const uint32_t px = bgr ? (lo | (hi << 8) | (hi << 16)) : (hi | (hi << 8) | (lo << 16));
    store_px_bytes(rgb + (size_t)i * rgb_pitch, j, px);
}

hipError_t launch_encode_depth(const float* depth, size_t depth_pitch, uint8_t* rgb, size_t rgb_pitch, int W, int H,
                               double max_depth, int bgr, hipStream_t s)
{
    dim3 grid((W + 255) / 256, H);
    hipLaunchKernelGGL(k_encode_depth, grid, dim3(256), 0, s, depth, depth_pitch, rgb, rgb_pitch, W, H,
                       4228250625.0 / max_depth, (float)max_depth, bgr);
    return hipGetLastError();
}

// =================================================================================================
// 89-degree oblique-triangle filter (dmt:1283-1294, 1339-1344), f64 exactly as NumPy >= 2 evaluates it
// =================================================================================================

__device__ __forceinline__ bool tri_oblique(const double (&a)[3], const double (&b)[3], const double (&c)[3])
{
    const double e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    const double e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    const double nx = e1y * e2z - e1z * e2y;
    const double ny = e1z * e2x - e1x * e2z;
    const double nz = e1x * e2y - e1y * e2x;
    const double vx = -((a[0] + b[0]) + c[0]) / 3.0;
    const double vy = -((a[1] + b[1]) + c[1]) / 3.0;
    const double vz = -((a[2] + b[2]) + c[2]) / 3.0;
    const double dot = (nx * vx + ny * vy) + nz * vz;
    const double len_n = sqrt((nx * nx + ny * ny) + nz * nz);
    const double len_v = sqrt((vx * vx + vy * vy) + vz * vz);
    const double cosine = dot / (len_n * len_v + 1e-15);
    return cosine < 0x1.1df0b2b89dd37p-6;          // np.cos(np.radians(89.0))
}

__device__ __forceinline__ void vertex_f64(const FrameDev& f, int i, int j, int of_by_one, float z, double (&p)[3])
{
    const double x = of_by_one ? (double)((float)j * f.sx) : (double)j;   // dmt:1117-1122 (f32 grid)
    const double y = of_by_one ? (double)((float)i * f.sy) : (double)i;
    p[0] = (x - f.Kd[2]) * (double)z / f.Kd[0];
    p[1] = (y - f.Kd[3]) * (double)z / f.Kd[1];
    p[2] = (double)z;
}

// One thread per grid cell: both triangles of the cell.  `unused` must be zeroed beforehand.
// NOTE f.sx / f.sy hold the mesh grid scale only when the frame was prepared for mesh mode; the host
// passes scale factors explicitly so the filter can be run standalone for either grid.
__global__ void k_edge_filter(const uint8_t* __restrict__ depth_rgb, size_t pitch, size_t stride,
                              const FrameDev* __restrict__ fp, int frame0, int W, int H, int of_by_one,
                              float sx, float sy,
                              uint8_t* __restrict__ tri_invalid, size_t tri_stride,
                              uint8_t* __restrict__ unused, size_t unused_stride)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int fr = blockIdx.z;
    if (j >= W - 1 || i >= H - 1) return;
    FrameDev f = fp[frame0 + fr];
    f.sx = sx; f.sy = sy;
    const uint8_t* r0 = depth_rgb + (size_t)(frame0 + fr) * stride + (size_t)i * pitch;
    const uint8_t* r1 = r0 + pitch;
    const float zA = decode_z(code16_of(load_px_bytes(r0, j)), f.mult, f.scale);
    const float zD = decode_z(code16_of(load_px_bytes(r0, j + 1)), f.mult, f.scale);
    const float zB = decode_z(code16_of(load_px_bytes(r1, j)), f.mult, f.scale);
    const float zC = decode_z(code16_of(load_px_bytes(r1, j + 1)), f.mult, f.scale);
    double A[3], B[3], Cc[3], D[3];
    vertex_f64(f, i, j, of_by_one, zA, A);
    vertex_f64(f, i + 1, j, of_by_one, zB, B);
    vertex_f64(f, i + 1, j + 1, of_by_one, zC, Cc);
    vertex_f64(f, i, j + 1, of_by_one, zD, D);
    const bool inv1 = tri_oblique(A, B, Cc);     // tri1 = (v[i,j], v[i+1,j], v[i+1,j+1])
    const bool inv2 = tri_oblique(A, Cc, D);     // tri2 = (v[i,j], v[i+1,j+1], v[i,j+1])
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    const size_t cell = (size_t)i * (W - 1) + j;
    if (tri_invalid) {
        uint8_t* t = tri_invalid + (size_t)fr * tri_stride;
        t[cell] = inv1;
        t[ncell + cell] = inv2;
    }
    if (unused && (inv1 || inv2)) {
        uint8_t* u = unused + (size_t)fr * unused_stride;
        const size_t a = (size_t)i * W + j;
        u[a] = 1;                      // A
        u[a + W + 1] = 1;              // C
        if (inv1) u[a + W] = 1;        // B
        if (inv2) u[a + 1] = 1;        // D
    }
}

hipError_t launch_edge_filter(const uint8_t* depth_rgb, size_t pitch, size_t stride, const FrameDev* fp, int frame0,
                              int n, int W, int H, int of_by_one, uint8_t* tri_invalid, size_t tri_stride,
                              uint8_t* unused, size_t unused_stride, hipStream_t s)
{
    const float sx = of_by_one ? (float)(((double)W + 1.0) / (double)W) : 1.0f;
    const float sy = of_by_one ? (float)(((double)H + 1.0) / (double)H) : 1.0f;
    dim3 grid((W - 1 + 127) / 128, H - 1, n);
    hipLaunchKernelGGL(k_edge_filter, grid, dim3(128), 0, s, depth_rgb, pitch, stride, fp, frame0, W, H, of_by_one,
                       sx, sy, tri_invalid, tri_stride, unused, unused_stride);
    return hipGetLastError();
}

// =================================================================================================
// POINT MODE, pure stereo shift: one workgroup per (frame,row), z-buffer in LDS
// =================================================================================================
//
// Row locality: with K == Krender, no pose and no toe-in, v = i exactly and u = j +- dl/Z, so an
// output row depends on one input row.  Z is strictly monotone in the 16-bit depth code, so the
// whole fragment fits one 64-bit LDS word
//        key = code16 << 40 | j << 24 | R | G<<8 | B<<16
// whose unsigned minimum is "nearest Z, ties to the lower source column" -- and it carries the
// colour, so the resolve phase is a plain LDS read (no gather, no second pass over the inputs).
// HBM traffic = algorithmic bytes: 6 B/px in, 8 B/px out (+8 B/px with the optional depth planes).

template <int PX>   // pixels per thread-iteration: 4 (dwordx3 path) or 1 (byte path, any W / alignment)
struct RowIO;

template <>
struct RowIO<4> {
    static __device__ __forceinline__ void load(const uint8_t* row, int g, uint32_t (&px)[4])
    {
        const uint32_t* p = (const uint32_t*)row + 3 * (size_t)g;
        unpack4(p[0], p[1], p[2], px);
    }
    static __device__ __forceinline__ void store_rgb(uint8_t* row, int g, const uint32_t (&px)[4])
    {
        uint32_t w0, w1, w2;
        pack4(px, w0, w1, w2);
        uint32_t* p = (uint32_t*)row + 3 * (size_t)g;
        p[0] = w0; p[1] = w1; p[2] = w2;
    }
    static __device__ __forceinline__ void store_mask(uint8_t* row, int g, const uint32_t (&m)[4])
    {
        ((uint32_t*)row)[g] = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
    }
    static __device__ __forceinline__ void store_z(float* row, int g, const float (&z)[4])
    {
        ((float4*)row)[g] = make_float4(z[0], z[1], z[2], z[3]);
    }
    static __device__ __forceinline__ void load_u8(const uint8_t* row, int g, uint32_t (&v)[4])
    {
        const uint32_t w = ((const uint32_t*)row)[g];
        v[0] = w & 0xFF; v[1] = (w >> 8) & 0xFF; v[2] = (w >> 16) & 0xFF; v[3] = w >> 24;
    }
};

template <>
struct RowIO<1> {
    static __device__ __forceinline__ void load(const uint8_t* row, int g, uint32_t (&px)[1]) { px[0] = load_px_bytes(row, g); }
    static __device__ __forceinline__ void store_rgb(uint8_t* row, int g, const uint32_t (&px)[1]) { store_px_bytes(row, g, px[0]); }
    static __device__ __forceinline__ void store_mask(uint8_t* row, int g, const uint32_t (&m)[1]) { row[g] = (uint8_t)m[0]; }
    static __device__ __forceinline__ void store_z(float* row, int g, const float (&z)[1]) { row[g] = z[0]; }
    static __device__ __forceinline__ void load_u8(const uint8_t* row, int g, uint32_t (&v)[1]) { v[0] = row[g]; }
};

// FLAGS bit 0: optional depth planes, bit 1: `unused` vertices are not drawn (remove_edges),
// bit 2: edge points splatted into holes.
template <int PX, int FLAGS>
__global__ void __launch_bounds__(256) k_points_rows(RenderArgs a)
{
    constexpr bool ZOUT = FLAGS & 1, UNUSED = FLAGS & 2, EDGE = FLAGS & 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int W = a.W;
    u64* zb = (u64*)smem;                        // [2][W] main z keys, left then right eye
    uint32_t* eb = (uint32_t*)(zb + 2 * (size_t)W);   // [2][W] edge-point keys (code16<<16 | j), EDGE only

    const int fr = blockIdx.x / a.H;
    const int i = blockIdx.x - fr * a.H;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const float mult = fp.mult, scale = fp.scale, dl = fp.dl;
    const int tid = threadIdx.x;
    const int ngroups = W / PX;

    for (int x = tid; x < 2 * W; x += blockDim.x) zb[x] = kEmpty64;
    if (EDGE) for (int x = tid; x < 2 * W; x += blockDim.x) eb[x] = kEmpty32;
    __syncthreads();

    const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch;
    const uint8_t* crow = a.color + (size_t)f * a.color_stride + (size_t)i * a.color_pitch;
    const uint8_t* urow = UNUSED ? a.unused + (size_t)fr * a.ws_stride_px + (size_t)i * W : nullptr;
    const float fW = (float)W;
    const float ecx = fp.cx, esW = fp.sW;

    for (int g = tid; g < ngroups; g += blockDim.x) {
        uint32_t dpx[PX], cpx[PX], un[PX];
        RowIO<PX>::load(drow, g, dpx);
        RowIO<PX>::load(crow, g, cpx);
        if (UNUSED) RowIO<PX>::load_u8(urow, g, un);
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int j = g * PX + q;
            const uint32_t code = code16_of(dpx[q]);
            const float z = decode_z(code, mult, scale);
            if (!(z > kNear)) continue;
            const float d = dl / z;
            const float fj = (float)j;
            if (!(UNUSED && un[q])) {
                const u64 key = ((u64)code << 40) | ((u64)(uint32_t)j << 24) | (u64)cpx[q];
                const float uL = fj + d, uR = fj - d;
                if (uL >= 0.0f && uL < fW) atomicMin(&zb[(int)floorf(uL)], key);
                if (uR >= 0.0f && uR < fW) atomicMin(&zb[W + (int)floorf(uR)], key);
            } else if (EDGE) {
                // sr:599-600, 746: undo the off-by-one scale on X, project, round half-even.
                const float ex = ((fj - ecx) * esW) + ecx;
                const uint32_t ekey = (code << 16) | (uint32_t)j;
                const float uL = ex + d, uR = ex - d;
                if (uL > -1.0f && uL < fW + 1.0f) {
                    const int x = (int)rintf(uL);
                    if (x >= 0 && x < W) atomicMin(&eb[x], ekey);
                }
                if (uR > -1.0f && uR < fW + 1.0f) {
                    const int x = (int)rintf(uR);
                    if (x >= 0 && x < W) atomicMin(&eb[W + x], ekey);
                }
            }
        }
    }
    __syncthreads();

#pragma unroll
    for (int eye = 0; eye < 2; ++eye) {
        uint8_t* orow = a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)i * a.rgb_pitch;
        uint8_t* mrow = a.mask[eye] + (size_t)f * a.mask_stride + (size_t)i * a.mask_pitch;
        float* zrow = ZOUT && a.zout[eye]
                          ? (float*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)i * a.zout_pitch)
                          : nullptr;
        const u64* zrow_lds = zb + (size_t)eye * W;
        for (int g = tid; g < ngroups; g += blockDim.x) {
            uint32_t opx[PX], om[PX];
            float oz[PX];
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const int x = g * PX + q;
                const u64 key = zrow_lds[x];
                const uint32_t rgb = (uint32_t)key & 0xFFFFFFu;
                const bool covered = key != kEmpty64;
                const bool hole = !covered || rgb == a.key_rgb;       // sr:740 colour-key compare
                uint32_t out = hole ? 0u : rgb;                       // sr:793
                if (EDGE && hole) {
                    const uint32_t ek = eb[(size_t)eye * W + x];
                    if (ek != kEmpty32) {
                        // colour of source column (ek & 0xFFFF) of this row (sr:813-814)
                        out = load_px_bytes(crow, (int)(ek & 0xFFFFu));
                    }
                }
                opx[q] = out;
                om[q] = hole ? 255u : 0u;
                if (ZOUT) oz[q] = covered ? decode_z((uint32_t)(key >> 40), mult, scale) : 0.0f;
            }
            RowIO<PX>::store_rgb(orow, g, opx);
            RowIO<PX>::store_mask(mrow, g, om);
            if (ZOUT && zrow) RowIO<PX>::store_z(zrow, g, oz);
        }
    }
}

// =================================================================================================
// POINT MODE, general (pose / convergence / K != Krender): global 64-bit z keys
// =================================================================================================
//   key = f32 bits of Z' << 32 | i << 16 | j      (Z' > 0, so the bit pattern orders like the value)

template <int FLAGS>
__global__ void __launch_bounds__(256) k_points_splat_general(RenderArgs a)
{
    constexpr bool UNUSED = FLAGS & 2, EDGE = FLAGS & 4;
    const int W = a.W, H = a.H;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    const int fr = blockIdx.z;
    if (j >= W) return;
    const int f = a.frame0 + fr;
    const FrameDev& fp = a.fp[f];
    const uint8_t* drow = a.depth + (size_t)f * a.depth_stride + (size_t)i * a.depth_pitch;
    const uint32_t code = code16_of(load_px_bytes(drow, j));
    const float z = decode_z(code, fp.mult, fp.scale);
    if (!(z > kNear)) return;
    const float gx = (float)j * fp.sx, gy = (float)i * fp.sy;
    float xc, yc;
    camera_point(fp, gx, gy, z, xc, yc);
    const bool un = UNUSED && a.unused[(size_t)fr * a.ws_stride_px + (size_t)i * W + j];
    const uint32_t src = ((uint32_t)i << 16) | (uint32_t)j;
    if (!un) {
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
            const Vert v = vertex_general(fp, fp.M[eye], xc, yc, z);
            if (!v.ok) continue;
            if (!(v.u >= 0.0f && v.u < (float)W && v.v >= 0.0f && v.v < (float)H)) continue;
            const int px = (int)floorf(v.u), py = (int)floorf(v.v);
            const u64 key = ((u64)__float_as_uint(v.z) << 32) | src;
            atomicMin(&a.keys[eye][(size_t)fr * a.ws_stride_px + (size_t)py * W + px], key);
        }
    } else if (EDGE) {
        const float xe = xc * fp.sW, ye = yc * fp.sH;        // sr:599-600
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
            const Vert v = vertex_general(fp, fp.M[eye], xe, ye, z);
            if (!v.ok) continue;
            if (!(v.u > -1.0f && v.u < (float)W + 1.0f && v.v > -1.0f && v.v < (float)H + 1.0f)) continue;
            const int px = (int)rintf(v.u), py = (int)rintf(v.v);   // np.round (sr:746)
            if (px < 0 || px >= W || py < 0 || py >= H) continue;
            const u64 key = ((u64)__float_as_uint(v.z) << 32) | src;
            atomicMin(&a.ekeys[eye][(size_t)fr * a.ws_stride_px + (size_t)py * W + px], key);
        }
    }
}

template <int FLAGS>
__global__ void __launch_bounds__(256) k_points_resolve_general(RenderArgs a)
{
    constexpr bool ZOUT = FLAGS & 1, EDGE = FLAGS & 4;
    const int W = a.W;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int fr = blockIdx.z >> 1, eye = blockIdx.z & 1;
    if (x >= W) return;
    const int f = a.frame0 + fr;
    const size_t o = (size_t)fr * a.ws_stride_px + (size_t)y * W + x;
    const u64 key = a.keys[eye][o];
    const bool covered = key != kEmpty64;
    uint32_t rgb = 0;
    const uint8_t* cbase = a.color + (size_t)f * a.color_stride;
    if (covered) {
        const uint32_t src = (uint32_t)key;
        rgb = load_px_bytes(cbase + (size_t)(src >> 16) * a.color_pitch, (int)(src & 0xFFFFu));
    }
    const bool hole = !covered || rgb == a.key_rgb;
    uint32_t out = hole ? 0u : rgb;
    if (EDGE && hole) {
        const u64 ek = a.ekeys[eye][o];
        if (ek != kEmpty64) {
            const uint32_t src = (uint32_t)ek;
            out = load_px_bytes(cbase + (size_t)(src >> 16) * a.color_pitch, (int)(src & 0xFFFFu));
        }
    }
    store_px_bytes(a.rgb[eye] + (size_t)f * a.rgb_stride + (size_t)y * a.rgb_pitch, x, out);
    (a.mask[eye] + (size_t)f * a.mask_stride + (size_t)y * a.mask_pitch)[x] = hole ? 255 : 0;
    if (ZOUT && a.zout[eye]) {
        float* zrow = (float*)((uint8_t*)a.zout[eye] + (size_t)f * a.zout_stride + (size_t)y * a.zout_pitch);
        zrow[x] = covered ? __uint_as_float((uint32_t)(key >> 32)) : 0.0f;
    }
}

// =================================================================================================
// launch plumbing
// =================================================================================================

size_t render_lds_bytes(const RenderPlan& plan, int W)
{
    if (plan.general) return 0;
    if (plan.mode == MDVT_MODE_POINTS) {
        size_t b = 2 * (size_t)W * sizeof(u64);
        if (plan.edge_points) b += 2 * (size_t)W * sizeof(uint32_t);
        return b;
    }
    return 0;
}

template <int PX>
static hipError_t launch_points_rows(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const size_t lds = render_lds_bytes(plan, a.W);
    const dim3 grid((unsigned)(plan.n * a.H)), block(256);
    const bool zout = a.zout[0] || a.zout[1];
    const int flags = (zout ? 1 : 0) | (plan.remove_edges ? 2 : 0) | (plan.remove_edges && plan.edge_points ? 4 : 0);
#define MDVT_CASE(F)                                                                                   \
    case F:                                                                                            \
        (void)hipFuncSetAttribute((const void*)k_points_rows<PX, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_points_rows<PX, F>), grid, block, lds, s, a);                            \
        break;
    switch (flags) {
        MDVT_CASE(0) MDVT_CASE(1) MDVT_CASE(2) MDVT_CASE(3) MDVT_CASE(6) MDVT_CASE(7)
        default: return hipErrorInvalidValue;
    }
#undef MDVT_CASE
    return hipGetLastError();
}

static hipError_t launch_points_general(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    const size_t npx = (size_t)a.W * a.H;
    const bool zout = a.zout[0] || a.zout[1];
    const bool edge = plan.remove_edges && plan.edge_points;
    hipError_t e;
    for (int eye = 0; eye < 2; ++eye) {
        if ((e = hipMemsetAsync(a.keys[eye], 0xFF, (size_t)plan.n * a.ws_stride_px * sizeof(u64), s)) != hipSuccess) return e;
        if (edge && (e = hipMemsetAsync(a.ekeys[eye], 0xFF, (size_t)plan.n * a.ws_stride_px * sizeof(u64), s)) != hipSuccess) return e;
    }
    (void)npx;
    const dim3 block(256);
    const dim3 grid_s((a.W + 255) / 256, a.H, plan.n);
    const dim3 grid_r((a.W + 255) / 256, a.H, plan.n * 2);
    const int sflags = (plan.remove_edges ? 2 : 0) | (edge ? 4 : 0);
    switch (sflags) {
        case 0: hipLaunchKernelGGL((k_points_splat_general<0>), grid_s, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_points_splat_general<2>), grid_s, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_points_splat_general<6>), grid_s, block, 0, s, a); break;
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    const int rflags = (zout ? 1 : 0) | (edge ? 4 : 0);
    switch (rflags) {
        case 0: hipLaunchKernelGGL((k_points_resolve_general<0>), grid_r, block, 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_points_resolve_general<1>), grid_r, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_points_resolve_general<4>), grid_r, block, 0, s, a); break;
        default: hipLaunchKernelGGL((k_points_resolve_general<5>), grid_r, block, 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_render(const RenderPlan& plan, const RenderArgs& a, hipStream_t s)
{
    if (plan.mode == MDVT_MODE_POINTS) {
        if (plan.general) return launch_points_general(plan, a, s);
        return plan.vec4 ? launch_points_rows<4>(plan, a, s) : launch_points_rows<1>(plan, a, s);
    }
    return hipErrorNotSupported;
}

}  // namespace mdvt
