/*
 * mdvt_oracle.c -- plain-C CPU restatement of the stereo-rerender hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mdvt_oracle.h).  It is written as the obvious sequential
 * algorithm -- full-frame z-buffers, painter's loop in draw order with a strict LESS depth
 * test -- and deliberately shares no code with the HIP kernels it checks.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC  (oracle/Makefile).
 * All f32 arithmetic below is written one operation per expression node; with contraction off
 * the compiler evaluates exactly the sequence DESIGN.md section "Arithmetic decree" lists.
 */
#include "mdvt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NEAR 1e-4f            /* ctr.set_constant_z_near(0.0001), dmt:1520 */
/* Raster sub-pixel grid: 2^subpixel_bits positions per pixel (GL_SUBPIXEL_BITS).  8 is the decree's default (what desktop
 * GPUs report); orc_params.subpixel_bits selects another grid so that the oracle can be held to a GL whose grid differs
 * (SwiftShader: 4, tests/golden/render_gl_*.npz).  The library is single threaded; the entry points set it per call. */
static int ORC_SUBPIX = 256;
static void orc_set_subpix(const orc_params* p)
{
    const int bits = (p->subpixel_bits >= 1 && p->subpixel_bits <= 8) ? p->subpixel_bits : 8;
    ORC_SUBPIX = 1 << bits;
}
#define ORC_SNAP_LIMIT 2097152.0f /* |u|,|v| clamp before snapping (2^21 px)   */

/* ------------------------------------------------------------------------------------------ */
/* codec                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* dfh:21-23: e.astype(float32) * (float(max_depth)/255**4); the python scalar is "weak", so the
 * multiply is f32 x f32 with the scalar rounded to f32 first. */
static float orc_decode_mult(double max_depth) { return (float)(max_depth / 4228250625.0); }

static inline float orc_decode_px(const uint8_t* px, float mult, float scale)
{
    /* dfh:67-69 (bit16): byte3 <- R, byte2 <- B, G ignored.  float(u32) is exact (16 sig. bits). */
    uint32_t code = ((uint32_t)px[0] << 24) | ((uint32_t)px[2] << 16);
    float d = (float)code * mult;
    return d * scale;                 /* sr:541  depth *= master_fov_scale_depth (f32 in place) */
}

void orc_decode_depth(const uint8_t* rgb, int W, int H, double max_depth, double depth_scale, float* out)
{
    const float mult = orc_decode_mult(max_depth);
    const float scale = (float)depth_scale;
    const size_t n = (size_t)W * (size_t)H;
    for (size_t i = 0; i < n; ++i) out[i] = orc_decode_px(rgb + 3 * i, mult, scale);
}

void orc_encode_depth(const float* depth, int W, int H, double max_depth, uint8_t* rgb)
{
    /* dfh:7-9: clip to [0,max] (f32 array clipped with python scalars -> stays f32), multiply in
     * f64 by 255**4/max_depth, truncate to uint32.  dfh:53-55: R = G = byte 3, B = byte 2. */
    const double multi = 4228250625.0 / max_depth;
    const float fmax = (float)max_depth;
    const size_t n = (size_t)W * (size_t)H;
    for (size_t i = 0; i < n; ++i) {
        float d = depth[i];
        if (d > fmax) d = fmax;       /* np.clip: min(max(d, 0), max); NaN propagates -> code 0 below */
        if (d < 0.0f) d = 0.0f;
        double e = multi * (double)d;
        uint32_t code = (e >= 0.0 && e < 4294967296.0) ? (uint32_t)e : 0u;
        rgb[3 * i + 0] = (uint8_t)(code >> 24);
        rgb[3 * i + 1] = (uint8_t)(code >> 24);
        rgb[3 * i + 2] = (uint8_t)((code >> 16) & 0xff);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* camera                                                                                     */
/* ------------------------------------------------------------------------------------------ */

int orc_camera_matrix(double xfov_deg, double yfov_deg, int W, int H, double* K9)
{
    const int hx = !isnan(xfov_deg), hy = !isnan(yfov_deg);
    double fx = 0.0, fy = 0.0;
    if (!hx && !hy) return -1;
    /* np.deg2rad(x) == x * (pi/180) */
    if (hx) fx = (double)W / (2.0 * tan((xfov_deg * (M_PI / 180.0)) / 2.0));
    if (hy) fy = (double)H / (2.0 * tan((yfov_deg * (M_PI / 180.0)) / 2.0));
    if (!hy) fy = fx;
    if (!hx) fx = fy;
    K9[0] = fx;  K9[1] = 0.0; K9[2] = (double)W / 2.0;
    K9[3] = 0.0; K9[4] = fy;  K9[5] = (double)H / 2.0;
    K9[6] = 0.0; K9[7] = 0.0; K9[8] = 1.0;
    return 0;
}

double orc_convergence_angle(double distance, double ipd_m) { return atan((ipd_m / 2.0) / distance); }

/* ------------------------------------------------------------------------------------------ */
/* f64 unprojection + edge filter (pinned by goldens from the reference)                      */
/* ------------------------------------------------------------------------------------------ */

static inline double orc_grid_f64(int idx, int n, int of_by_one)
{
    /* dmt:1117-1122: x.astype(float32); x *= (width+1)/width  (f32 array times weak python float) */
    if (!of_by_one) return (double)idx;
    const float s = (float)(((double)n + 1.0) / (double)n);
    return (double)((float)idx * s);
}

void orc_unproject_f64(const float* depth, int W, int H, const double* K4, int of_by_one, double* out)
{
    const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    for (int i = 0; i < H; ++i) {
        const double y = orc_grid_f64(i, H, of_by_one);
        for (int j = 0; j < W; ++j) {
            const double x = orc_grid_f64(j, W, of_by_one);
            const double z = (double)depth[(size_t)i * W + j];
            double* o = out + 3 * ((size_t)i * W + j);
            o[0] = (x - cx) * z / fx;          /* dmt:1127 */
            o[1] = (y - cy) * z / fy;          /* dmt:1128 */
            o[2] = z;
        }
    }
}

/* cos(radians(89.0)) as NumPy evaluates it (dmt:1287); pinned by tests/golden. */
static const double ORC_COS_89 = 0x1.1df0b2b89dd37p-6;

static int orc_tri_invalid(const double* a, const double* b, const double* c, double* nrm)
{
    /* dmt:1283-1294 */
    const double e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] };
    const double e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
    const double n[3] = { e1[1] * e2[2] - e1[2] * e2[1],
                          e1[2] * e2[0] - e1[0] * e2[2],
                          e1[0] * e2[1] - e1[1] * e2[0] };
    const double v[3] = { -((a[0] + b[0]) + c[0]) / 3.0,
                          -((a[1] + b[1]) + c[1]) / 3.0,
                          -((a[2] + b[2]) + c[2]) / 3.0 };
    const double dot = (n[0] * v[0] + n[1] * v[1]) + n[2] * v[2];
    const double len_n = sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    const double len_v = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    const double cosine = dot / (len_n * len_v + 1e-15);
    if (nrm) {
        /* dmt:1346-1353: area2 = np.linalg.norm(normals, axis=1) -> sqrt of the sum of squares;
         * unit normal, or (1,1,1) where the triangle is degenerate. */
        if (len_n > 0.0) { nrm[0] = n[0] / len_n; nrm[1] = n[1] / len_n; nrm[2] = n[2] / len_n; }
        else { nrm[0] = nrm[1] = nrm[2] = 1.0; }
    }
    return cosine < ORC_COS_89;
}

void orc_edge_filter(const float* depth, int W, int H, const double* K4, int of_by_one,
                     uint8_t* tri_invalid, uint8_t* unused, double* normals)
{
    const size_t nv = (size_t)W * H;
    const size_t ncell = (size_t)(W - 1) * (H - 1);
    double* P = (double*)malloc(nv * 3 * sizeof(double));
    orc_unproject_f64(depth, W, H, K4, of_by_one, P);
    memset(unused, 0, nv);
    if (normals) memset(normals, 0, nv * 3 * sizeof(double));
    /* draw order (dmt:1243-1254): all tri1 = (v[i,j], v[i+1,j], v[i+1,j+1]) row-major over cells,
     * then all tri2 = (v[i,j], v[i+1,j+1], v[i,j+1]). */
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < H - 1; ++i) {
            for (int j = 0; j < W - 1; ++j) {
                const size_t i1 = (size_t)i * W + j, i2 = (size_t)(i + 1) * W + j;
                const size_t i3 = (size_t)(i + 1) * W + j + 1, i4 = (size_t)i * W + j + 1;
                const size_t v0 = i1, v1 = pass == 0 ? i2 : i3, v2 = pass == 0 ? i3 : i4;
                double nrm[3];
                const int inv = orc_tri_invalid(P + 3 * v0, P + 3 * v1, P + 3 * v2, normals ? nrm : NULL);
                tri_invalid[(size_t)pass * ncell + (size_t)i * (W - 1) + j] = (uint8_t)inv;
                if (inv) { unused[v0] = 1; unused[v1] = 1; unused[v2] = 1; }   /* dmt:1339-1344 */
                if (normals) {                                                  /* dmt:1358-1364 */
                    memcpy(normals + 3 * v0, nrm, sizeof nrm);
                    memcpy(normals + 3 * v1, nrm, sizeof nrm);
                    memcpy(normals + 3 * v2, nrm, sizeof nrm);
                }
            }
        }
    }
    free(P);
}

/* ------------------------------------------------------------------------------------------ */
/* per-eye vertex programme (f32, decree)                                                     */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    int W, H;
    int general;
    int of_by_one;
    float mult, scale;          /* codec */
    float fx, fy, cx, cy;       /* input camera, f32 */
    float fxr, fyr, cxr, cyr;   /* render camera, f32 */
    float sx, sy;               /* (W+1)/W, (H+1)/H in f32 (or 1) */
    float dl;                   /* fxr * ipd/2 in f32: pure-shift disparity numerator */
    float sign;                 /* +1 left eye, -1 right eye (pure shift) */
    float M[12];                /* general: 3x4 eye*pose matrix, f32 */
} orc_eye;

static void orc_eye_setup(const orc_params* p, int eye /*0 left, 1 right*/, orc_eye* e)
{
    memset(e, 0, sizeof *e);
    e->W = p->W; e->H = p->H;
    e->general = p->general;
    e->of_by_one = (p->mode == ORC_MODE_MESH);        /* sr:574-580 */
    e->mult = orc_decode_mult(p->max_depth);
    e->scale = (float)p->depth_scale;
    e->fx = (float)p->K[0];  e->fy = (float)p->K[1];  e->cx = (float)p->K[2];  e->cy = (float)p->K[3];
    e->fxr = (float)p->Kr[0]; e->fyr = (float)p->Kr[1]; e->cxr = (float)p->Kr[2]; e->cyr = (float)p->Kr[3];
    e->sx = e->of_by_one ? (float)(((double)p->W + 1.0) / (double)p->W) : 1.0f;
    e->sy = e->of_by_one ? (float)(((double)p->H + 1.0) / (double)p->H) : 1.0f;
    const double half = p->ipd_m / 2.0;
    e->dl = (float)(p->Kr[0] * half);
    e->sign = eye == 0 ? 1.0f : -1.0f;
    /* general: M = Translate(+-ipd/2) * Ry(-+a) * T   (sr:615-619, 724-725, 832-836).
     * Ry(t) = [[c,0,s],[0,1,0],[-s,0,c]] (Open3D get_rotation_matrix_from_xyz((0,t,0))). */
    const double t = eye == 0 ? -p->conv_angle : p->conv_angle;
    const double c = cos(t), s = sin(t);
    const double R[3][3] = { { c, 0.0, s }, { 0.0, 1.0, 0.0 }, { -s, 0.0, c } };
    double T[16] = { 1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1 };
    if (p->has_T) memcpy(T, p->T, sizeof T);
    const double shift[3] = { eye == 0 ? half : -half, 0.0, 0.0 };
    for (int r = 0; r < 3; ++r) {
        for (int col = 0; col < 3; ++col)
            e->M[4 * r + col] = (float)((R[r][0] * T[0 + col] + R[r][1] * T[4 + col]) + R[r][2] * T[8 + col]);
        e->M[4 * r + 3] = (float)(((R[r][0] * T[3] + R[r][1] * T[7]) + R[r][2] * T[11]) + shift[r]);
    }
}

typedef struct { float u, v, z; int ok; } orc_vert;

/* Mesh / point vertex (i,j) with source depth zsrc -> screen position + eye-space depth. */
static inline orc_vert orc_vertex(const orc_eye* e, int i, int j, float zsrc)
{
    orc_vert o;
    const float gx = (float)j * e->sx;
    const float gy = (float)i * e->sy;
    if (!e->general) {
        /* pure shift: u = grid_x + fx*s/Z, v = grid_y (exact-arithmetic form of dmt:1127-1128
         * followed by the +-ipd/2 translate and the pinhole projection with K == Krender). */
        const float d = e->dl / zsrc;
        o.u = e->sign > 0.0f ? gx + d : gx - d;
        o.v = gy;
        o.z = zsrc;
        o.ok = zsrc > ORC_NEAR;
        return o;
    }
    const float xc = ((gx - e->cx) * zsrc) / e->fx;
    const float yc = ((gy - e->cy) * zsrc) / e->fy;
    const float* M = e->M;
    const float X = ((M[0] * xc + M[1] * yc) + M[2] * zsrc) + M[3];
    const float Y = ((M[4] * xc + M[5] * yc) + M[6] * zsrc) + M[7];
    const float Z = ((M[8] * xc + M[9] * yc) + M[10] * zsrc) + M[11];
    o.ok = (zsrc > ORC_NEAR) && (Z > ORC_NEAR);
    o.u = (e->fxr * X) / Z + e->cxr;
    o.v = (e->fyr * Y) / Z + e->cyr;
    o.z = Z;
    return o;
}

/* ------------------------------------------------------------------------------------------ */
/* edge points: the reference's own f64 chain (sr:589-606, 615-619, 727-735, 745-752, 838-858)  */
/* ------------------------------------------------------------------------------------------ */
/* A vertex of a removed triangle is splatted where THIS sequence of f64 operations puts it -- NumPy's unprojection
 * (dmt:1117-1128, f64 under NumPy >= 2), the "undo" of the off-by-one scale (sr:599-600), Open3D's in-place
 * transform / rotate / translate of the point cloud (Geometry3D::TransformPoints, RotatePoints, TranslatePoints:
 * 4x4 times (x,y,z,1) divided by w; R (p - 0) + 0; p += t), the right eye's operations applied ON TOP of the left
 * eye's (sr:838-847), cv2.projectPoints with the camera matrix cast to f32 (dmt:1058; cvProjectPoints2 converts
 * everything to double: z = z ? 1/z : 1, x *= z, u = x fx + cx), np.round (sr:746, 858).  One IEEE operation per
 * node, sums left to right, no contraction.  Pinned by tests/golden/edge_points.npz (the loop body's statements run
 * on the reference's own functions).  Exact simplifications used below: a product with an exact 0 or 1 entry of
 * Ry / of an affine pose's last row, and the additions of 0.0 in y and z of a translate, change no finite value. */
typedef struct {
    int W, H, of_by_one;
    float sx, sy;
    double fx, fy, cx, cy;          /* input camera (f64) */
    double fxr, fyr, cxr, cyr;      /* render camera, each rounded to f32 first (dmt:1058) */
    double sW, sH, h;
    int has_T, has_conv;
    double T[16], c, s;
} orc_echain;

static void orc_echain_setup(const orc_params* p, orc_echain* e)
{
    memset(e, 0, sizeof *e);
    e->W = p->W; e->H = p->H;
    e->of_by_one = (p->mode == ORC_MODE_MESH);
    e->sx = e->of_by_one ? (float)(((double)p->W + 1.0) / (double)p->W) : 1.0f;
    e->sy = e->of_by_one ? (float)(((double)p->H + 1.0) / (double)p->H) : 1.0f;
    e->fx = p->K[0]; e->fy = p->K[1]; e->cx = p->K[2]; e->cy = p->K[3];
    e->fxr = (double)(float)p->Kr[0]; e->fyr = (double)(float)p->Kr[1];
    e->cxr = (double)(float)p->Kr[2]; e->cyr = (double)(float)p->Kr[3];
    e->sW = ((double)p->W - 1.0) / (double)p->W;
    e->sH = ((double)p->H - 1.0) / (double)p->H;
    e->h = p->ipd_m / 2.0;
    e->has_T = p->has_T;
    if (p->has_T) memcpy(e->T, p->T, sizeof e->T);
    e->has_conv = (p->conv_angle == p->conv_angle) && p->conv_angle != 0.0;
    e->c = cos(p->conv_angle); e->s = sin(p->conv_angle);
}

/* dmt:1117-1128: the vertex as NumPy leaves it in mesh.vertices */
static inline void orc_echain_vertex(const orc_echain* e, int i, int j, float zsrc, double* P)
{
    const double gx = e->of_by_one ? (double)((float)j * e->sx) : (double)j;
    const double gy = e->of_by_one ? (double)((float)i * e->sy) : (double)i;
    P[0] = ((gx - e->cx) * (double)zsrc) / e->fx;
    P[1] = ((gy - e->cy) * (double)zsrc) / e->fy;
    P[2] = (double)zsrc;
}

static inline void orc_echain_roty(double c, double s, double* q)     /* Ry = [[c,0,s],[0,1,0],[-s,0,c]] */
{
    const double x = c * q[0] + s * q[2];
    const double z = (-s) * q[0] + c * q[2];
    q[0] = x; q[2] = z;
}

/* a point of the cloud (already scaled / offset as the caller needs) -> where it stands for the left and the right eye */
static void orc_echain_eyes(const orc_echain* e, const double* q0, double* L, double* R)
{
    double q[3] = { q0[0], q0[1], q0[2] };
    if (e->has_T) {                                                  /* sr:615-619 */
        const double* T = e->T;
        double hh[4];
        for (int r = 0; r < 4; ++r) hh[r] = ((T[4 * r] * q[0] + T[4 * r + 1] * q[1]) + T[4 * r + 2] * q[2]) + T[4 * r + 3] * 1.0;
        q[0] = hh[0] / hh[3]; q[1] = hh[1] / hh[3]; q[2] = hh[2] / hh[3];
    }
    if (e->has_conv) orc_echain_roty(e->c, -e->s, q);                /* sr:729 */
    q[0] += e->h;                                                    /* sr:731: translate([-left_shift, 0, 0]) */
    L[0] = q[0]; L[1] = q[1]; L[2] = q[2];
    q[0] += -e->h;                                                   /* sr:839 */
    if (e->has_conv) { orc_echain_roty(e->c, e->s, q); orc_echain_roty(e->c, e->s, q); }   /* sr:842-843 */
    q[0] += -e->h;                                                   /* sr:846 */
    R[0] = q[0]; R[1] = q[1]; R[2] = q[2];
}

/* cv2.projectPoints + np.round; returns 0 if the rounded pixel lies outside the frame (sr:747-750) */
static inline int orc_echain_pixel(const orc_echain* e, const double* q, int* px, int* py)
{
    const double iz = q[2] != 0.0 ? 1.0 / q[2] : 1.0;
    const double u = (q[0] * iz) * e->fxr + e->cxr;
    const double v = (q[1] * iz) * e->fyr + e->cyr;
    const double ru = rint(u), rv = rint(v);
    if (!(ru >= 0.0 && ru < (double)e->W && rv >= 0.0 && rv < (double)e->H)) return 0;     /* (NaN fails) */
    *px = (int)ru; *py = (int)rv;
    return 1;
}

/* Edge point of vertex (i, j) for one eye: pixel, and the depth the painter's order sorts by (sr:752) rounded to f32
 * (decree: the nearest point wins, an exact tie in that f32 goes to the lower source index -- the reference's
 * unstable argsort leaves ties undefined).  Vertices of depth code 0 (Z = 0: the reference would splat them all onto
 * the one pixel round(+-ipd/2 fx + cx, cy)) are not splatted (decree, unchanged). */
static inline int orc_edge_point(const orc_echain* e, int eye, int i, int j, float zsrc, int* px, int* py, float* zout)
{
    double P[3], L[3], R[3];
    if (!(zsrc > ORC_NEAR)) return 0;
    orc_echain_vertex(e, i, j, zsrc, P);
    P[0] *= e->sW; P[1] *= e->sH;                                    /* sr:599-600 */
    orc_echain_eyes(e, P, L, R);
    const double* q = eye == 0 ? L : R;
    if (!orc_echain_pixel(e, q, px, py)) return 0;
    *zout = (float)q[2];
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* rasteriser (decree: 1/256 sub-pixel snap, integer edge functions, top-left rule, no culling) */
/* ------------------------------------------------------------------------------------------ */

static inline int64_t orc_snap(float x)
{
    if (x > ORC_SNAP_LIMIT) x = ORC_SNAP_LIMIT;
    if (x < -ORC_SNAP_LIMIT) x = -ORC_SNAP_LIMIT;
    return (int64_t)rintf(x * (float)ORC_SUBPIX);
}

static inline int64_t orc_floordiv(int64_t a, int64_t b) /* b > 0 */
{
    int64_t q = a / b;
    if ((a % b) < 0) --q;
    return q;
}

typedef struct {
    int W, H;
    float* zbuf;        /* mesh: interpolated 1/Z (bigger = nearer), 0 = empty; points: Z, INF = empty */
    float* zinv;        /* mesh: 1/zbuf of the winning fragment (the depth plane) */
    uint8_t* rgb;       /* H*W*3 */
    uint8_t* covered;   /* H*W   */
} orc_target;

static inline int orc_edge_in(int64_t w, int64_t dx, int64_t dy)
{
    if (w > 0) return 1;
    if (w < 0) return 0;
    /* A pixel centre exactly on an edge belongs to the triangle if the edge is a LEFT edge or a horizontal BOTTOM edge (image
     * space, y down; orientation normalised to clockwise).  This is OpenGL's fill convention: the hardware's "top-left" rule
     * acts towards window y = 0, which is the BOTTOM of the picture (observed on the pinned GL: tests/golden/render_gl_*.npz;
     * Direct3D, whose y = 0 is the top, gives the same edges the name top-left). */
    return (dy < 0) || (dy == 0 && dx < 0);
}

static int64_t g_stats[5];
void orc_stats_reset(void) { memset(g_stats, 0, sizeof g_stats); }
void orc_stats_get(int64_t out[5]) { memcpy(out, g_stats, sizeof g_stats); }

static void orc_raster_tri(orc_target* t, const orc_vert* a, const orc_vert* b, const orc_vert* c,
                           const uint8_t* ca, const uint8_t* cb, const uint8_t* cc, int cull)
{
    if (!(a->ok && b->ok && c->ok)) {           /* near-plane: drop the whole triangle (decree) */
        g_stats[(a->ok || b->ok || c->ok) ? 0 : 1]++;
        return;
    }
    const int64_t X0 = orc_snap(a->u), Y0 = orc_snap(a->v);
    const int64_t X1 = orc_snap(b->u), Y1 = orc_snap(b->v);
    const int64_t X2 = orc_snap(c->u), Y2 = orc_snap(c->v);
    int64_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
    if (area2 == 0) return;
    /* y runs down: the grid's own triangles (dmt:1243-1254) have negative area2 here and appear counter-clockwise on
     * screen -- GL's default front face */
    if ((cull == 1 && area2 > 0) || (cull == 2 && area2 < 0)) { g_stats[4]++; return; }
    const int64_t s = area2 > 0 ? 1 : -1;
    area2 *= s;
    int64_t minX = X0 < X1 ? X0 : X1; if (X2 < minX) minX = X2;
    int64_t maxX = X0 > X1 ? X0 : X1; if (X2 > maxX) maxX = X2;
    int64_t minY = Y0 < Y1 ? Y0 : Y1; if (Y2 < minY) minY = Y2;
    int64_t maxY = Y0 > Y1 ? Y0 : Y1; if (Y2 > maxY) maxY = Y2;
    const int64_t half = ORC_SUBPIX / 2;
    int64_t px0 = orc_floordiv(minX - half + ORC_SUBPIX - 1, ORC_SUBPIX);   /* ceil((minX-128)/256) */
    int64_t px1 = orc_floordiv(maxX - half, ORC_SUBPIX);
    int64_t py0 = orc_floordiv(minY - half + ORC_SUBPIX - 1, ORC_SUBPIX);
    int64_t py1 = orc_floordiv(maxY - half, ORC_SUBPIX);
    if (px0 < 0) px0 = 0;
    if (py0 < 0) py0 = 0;
    if (px1 > t->W - 1) px1 = t->W - 1;
    if (py1 > t->H - 1) py1 = t->H - 1;
    /* directed edges opposite each vertex, orientation-normalised */
    const int64_t dx0 = s * (X2 - X1), dy0 = s * (Y2 - Y1);   /* v1 -> v2, weight of v0 */
    const int64_t dx1 = s * (X0 - X2), dy1 = s * (Y0 - Y2);   /* v2 -> v0, weight of v1 */
    const int64_t dx2 = s * (X1 - X0), dy2 = s * (Y1 - Y0);   /* v0 -> v1, weight of v2 */
    const float iz0 = 1.0f / a->z, iz1 = 1.0f / b->z, iz2 = 1.0f / c->z;
    const float ra = 1.0f / (float)area2;        /* one division per triangle; lambda_k = f32(w_k) * ra */
    for (int64_t py = py0; py <= py1; ++py) {
        const int64_t Yc = py * ORC_SUBPIX + half;
        for (int64_t px = px0; px <= px1; ++px) {
            const int64_t Xc = px * ORC_SUBPIX + half;
            const int64_t w0 = s * ((X2 - X1) * (Yc - Y1) - (Y2 - Y1) * (Xc - X1));
            const int64_t w1 = s * ((X0 - X2) * (Yc - Y2) - (Y0 - Y2) * (Xc - X2));
            const int64_t w2 = s * ((X1 - X0) * (Yc - Y0) - (Y1 - Y0) * (Xc - X0));
            if (!(orc_edge_in(w0, dx0, dy0) && orc_edge_in(w1, dx1, dy1) && orc_edge_in(w2, dx2, dy2)))
                continue;
            const float l0 = (float)w0 * ra, l1 = (float)w1 * ra, l2 = (float)w2 * ra;
            const float q0 = l0 * iz0, q1 = l1 * iz1, q2 = l2 * iz2;
            const float iz = (q0 + q1) + q2;
            const size_t o = (size_t)py * t->W + (size_t)px;
            g_stats[3]++;
            if (t->covered[o] && iz == t->zbuf[o]) g_stats[2]++;
            /* GL_LESS on depth == strictly GREATER on 1/Z: on an exact tie the triangle drawn first keeps the pixel
             * (draw order = all tri1 row-major, then all tri2, dmt:1243-1254) */
            if (t->covered[o] && !(iz > t->zbuf[o])) continue;
            const float riz = 1.0f / iz;               /* one division per fragment; also the depth plane value */
            uint8_t frag[3];
            for (int ch = 0; ch < 3; ++ch) {
                const float num = (q0 * (float)ca[ch] + q1 * (float)cb[ch]) + q2 * (float)cc[ch];
                float val = rintf(num * riz);
                if (!(val >= 0.0f)) val = 0.0f;
                if (val > 255.0f) val = 255.0f;
                frag[ch] = (uint8_t)val;
            }
            t->zbuf[o] = iz;
            t->covered[o] = 1;
            t->zinv[o] = riz;
            memcpy(t->rgb + 3 * o, frag, 3);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* one eye                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* Normal colour of an edge point for the infill-mask seed image (sr:596-606, 727-733, 777-802): the removed
 * vertex normal n is carried as the point n + p (p BEFORE the undo scale, sr:596), both it and the undo-scaled edge
 * point go through the same chain of operations, their difference is normalised and stored as (n'+1)/2 * 255, truncated. */
static void orc_edge_normal_colour(const orc_echain* e, int eye, const double* p, const double* n, uint8_t* rgb)
{
    const double a[3] = { n[0] + p[0], n[1] + p[1], n[2] + p[2] };
    const double q[3] = { p[0] * e->sW, p[1] * e->sH, p[2] };
    double aL[3], aR[3], qL[3], qR[3], d[3];
    orc_echain_eyes(e, a, aL, aR);
    orc_echain_eyes(e, q, qL, qR);
    for (int r = 0; r < 3; ++r) d[r] = eye == 0 ? aL[r] - qL[r] : aR[r] - qR[r];
    const double len = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (int r = 0; r < 3; ++r) {
        const double c = ((d[r] / len) + 1.0) / 2.0 * 255.0;
        rgb[r] = (c >= 0.0 && c < 256.0) ? (uint8_t)c : 0;            /* NaN (zero-length) -> 0 */
    }
}

static void orc_render_eye(const orc_params* p, int eye, const float* depth, const uint8_t* color,
                           const uint8_t* tri_invalid, const uint8_t* unused,
                           uint8_t* out_rgb, uint8_t* out_mask, float* out_depth,
                           const double* P, const double* vnormals, uint8_t* out_seed)
{
    const int W = p->W, H = p->H;
    const size_t n = (size_t)W * H;
    orc_eye e;
    orc_eye_setup(p, eye, &e);

    orc_target t;
    t.W = W; t.H = H;
    t.zbuf = (float*)malloc(n * sizeof(float));
    t.zinv = (float*)calloc(n, sizeof(float));
    t.rgb = (uint8_t*)calloc(n, 3);
    t.covered = (uint8_t*)calloc(n, 1);

    orc_vert* V = (orc_vert*)malloc(n * sizeof(orc_vert));
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j)
            V[(size_t)i * W + j] = orc_vertex(&e, i, j, depth[(size_t)i * W + j]);

    if (p->mode == ORC_MODE_POINTS) {
        /* GL_POINTS of size 1 (dmt:1510): pixel = (floor u, floor v); nearest Z wins; ties go to
         * the lower source index (GL_LESS + draw order).  Vertices of removed triangles are parked
         * behind the camera (dmt:1091) == not drawn. */
        for (size_t k = 0; k < n; ++k) t.zbuf[k] = INFINITY;
        for (size_t k = 0; k < n; ++k) {
            const orc_vert* v = &V[k];
            if (!v->ok) continue;
            if (unused && unused[k]) continue;
            /* A size-1 point is the unit square around the SNAPPED vertex, rasterised like any polygon with the fill rule of
             * orc_edge_in: the pixel whose centre lies inside it, a centre on the square's left or bottom edge belongs to it,
             * on its right or top edge not:  px = ceil(X/S) - 1, py = floor(Y/S), with X, Y the snapped position in 1/S
             * pixels.  (Observed on the pinned GL, tests/golden/render_gl_*.npz: a point at u in [j, j + 1/(2S)] lands in
             * column j - 1.) */
            const int64_t X = orc_snap(v->u), Y = orc_snap(v->v);
            const int64_t px64 = orc_floordiv(X - 1, ORC_SUBPIX), py64 = orc_floordiv(Y, ORC_SUBPIX);
            if (px64 < 0 || px64 >= W || py64 < 0 || py64 >= H) continue;
            const int px = (int)px64, py = (int)py64;
            const size_t o = (size_t)py * W + px;
            if (!(v->z < t.zbuf[o])) continue;
            t.zbuf[o] = v->z;
            t.covered[o] = 1;
            memcpy(t.rgb + 3 * o, color + 3 * k, 3);
        }
        if (out_depth)
            for (size_t k = 0; k < n; ++k) out_depth[k] = t.covered[k] ? t.zbuf[k] : 0.0f;
    } else {
        for (size_t k = 0; k < n; ++k) t.zbuf[k] = 0.0f;
        const size_t ncell = (size_t)(W - 1) * (H - 1);
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < H - 1; ++i)
                for (int j = 0; j < W - 1; ++j) {
                    if (tri_invalid && tri_invalid[(size_t)pass * ncell + (size_t)i * (W - 1) + j])
                        continue;                            /* dmt:1372: zeroed == draws nothing */
                    const size_t i1 = (size_t)i * W + j, i2 = (size_t)(i + 1) * W + j;
                    const size_t i3 = (size_t)(i + 1) * W + j + 1, i4 = (size_t)i * W + j + 1;
                    const size_t v0 = i1, v1 = pass == 0 ? i2 : i3, v2 = pass == 0 ? i3 : i4;
                    orc_raster_tri(&t, &V[v0], &V[v1], &V[v2], color + 3 * v0, color + 3 * v1, color + 3 * v2, p->cull);
                }
        if (out_depth)
            for (size_t k = 0; k < n; ++k) out_depth[k] = t.covered[k] ? t.zinv[k] : 0.0f;
    }

    /* hole mask (sr:740): colour-key compare against the background colour. */
    for (size_t k = 0; k < n; ++k) {
        const uint8_t* c = t.rgb + 3 * k;
        const int hole = !t.covered[k] ||
                         (c[0] == p->key_rgb[0] && c[1] == p->key_rgb[1] && c[2] == p->key_rgb[2]);
        out_mask[k] = hole ? 255 : 0;
        if (hole) { out_rgb[3 * k] = out_rgb[3 * k + 1] = out_rgb[3 * k + 2] = 0; }   /* sr:793 */
        else memcpy(out_rgb + 3 * k, c, 3);
    }

    /* infill-mask seed (sr:787-799), before the edge normals: key colour in holes, black elsewhere, fixed
     * inward normals on hole pixels of the image border (columns first, then rows). */
    if (out_seed) {
        for (size_t k = 0; k < n; ++k) {
            uint8_t* s3 = out_seed + 3 * k;
            if (!out_mask[k]) { s3[0] = s3[1] = s3[2] = 0; continue; }
            const int x = (int)(k % (size_t)W), y = (int)(k / (size_t)W);
            if (x == 0) { s3[0] = 255; s3[1] = 127; s3[2] = 127; }
            else if (x == W - 1) { s3[0] = 0; s3[1] = 127; s3[2] = 127; }
            else if (y == 0) { s3[0] = 127; s3[1] = 127; s3[2] = 0; }
            else if (y == H - 1) { s3[0] = 127; s3[1] = 127; s3[2] = 255; }
            else memcpy(s3, p->key_rgb, 3);
        }
    }

    /* edge points (sr:745-781, 813-814): far-to-near overwrite == nearest wins, only into holes. */
    if (p->edge_points && unused) {
        orc_echain ec;
        orc_echain_setup(p, &ec);
        float* ez = (float*)malloc(n * sizeof(float));
        for (size_t k = 0; k < n; ++k) ez[k] = INFINITY;
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                const size_t k = (size_t)i * W + j;
                int px, py; float z;
                if (!unused[k]) continue;
                if (!orc_edge_point(&ec, eye, i, j, depth[k], &px, &py, &z)) continue;
                const size_t o = (size_t)py * W + px;
                if (!out_mask[o]) continue;                  /* sr:776: only where still background */
                if (!(z < ez[o])) continue;                  /* ties: lower source index wins (decree) */
                ez[o] = z;
                if (p->edge_points != 2) memcpy(out_rgb + 3 * o, color + 3 * k, 3);      /* 2: seed only (sr:809-812) */
                if (out_seed) orc_edge_normal_colour(&ec, eye, P + 3 * k, vnormals + 3 * k, out_seed + 3 * o);   /* sr:802 */
            }
        free(ez);
    }

    free(V); free(t.covered); free(t.rgb); free(t.zinv); free(t.zbuf);
}

int orc_render_stereo_seed(const orc_params* p, const uint8_t* depth_rgb, const uint8_t* color_rgb,
                           uint8_t* left_rgb, uint8_t* right_rgb, uint8_t* left_mask, uint8_t* right_mask,
                           float* left_depth, float* right_depth, uint8_t* left_seed, uint8_t* right_seed)
{
    if (!p || p->W < 2 || p->H < 2) return -1;
    if (p->mode != ORC_MODE_POINTS && p->mode != ORC_MODE_MESH) return -1;
    const int W = p->W, H = p->H;
    const size_t n = (size_t)W * H;
    const int want_seed = left_seed || right_seed;
    orc_set_subpix(p);
    float* depth = (float*)malloc(n * sizeof(float));
    orc_decode_depth(depth_rgb, W, H, p->max_depth, p->depth_scale, depth);

    uint8_t* tri_invalid = NULL; uint8_t* unused = NULL;
    double* P = NULL; double* vn = NULL;
    if (p->remove_edges) {
        tri_invalid = (uint8_t*)malloc(2 * (size_t)(W - 1) * (H - 1));
        unused = (uint8_t*)malloc(n);
        if (want_seed) {
            vn = (double*)malloc(n * 3 * sizeof(double));
            P = (double*)malloc(n * 3 * sizeof(double));
            orc_unproject_f64(depth, W, H, p->K, p->mode == ORC_MODE_MESH, P);
        }
        orc_edge_filter(depth, W, H, p->K, p->mode == ORC_MODE_MESH, tri_invalid, unused, vn);
    }
    orc_render_eye(p, 0, depth, color_rgb, tri_invalid, unused, left_rgb, left_mask, left_depth, P, vn, left_seed);
    orc_render_eye(p, 1, depth, color_rgb, tri_invalid, unused, right_rgb, right_mask, right_depth, P, vn, right_seed);
    free(tri_invalid); free(unused); free(depth); free(P); free(vn);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* GL candidates (diagnostics): what the states dmt.render leaves to the GL would change        */
/* ------------------------------------------------------------------------------------------ */
/* The decree above fixes three things a real OpenGL does differently or leaves to the implementation:
 *   near_clip      GL CLIPS a triangle that straddles the near plane (dmt:1520: 1e-4); the decree drops it whole.
 *   samples = 4    Open3D's legacy window asks for a 4x multisampled framebuffer; the decree samples once, at the centre.
 *   depth_tie_tol  GL compares WINDOW depth, 24 bits (or an f32 just below 1.0) of 1 - near/Z: with near = 1e-4 two
 *                  fragments closer than ~6e-4 Z^2 metres are the same depth (GL_LESS: the first drawn stays) or differ by
 *                  the implementation's rounding noise; the decree compares f32 1/Z exactly.
 * orc_render_stereo_gl renders with any of them switched on, with the SAME vertex programme, snap, fill rule, draw order
 * and shading arithmetic as the decree (with every option off it is held bit for bit to orc_render_stereo by
 * tests/test_oracle_golden.py), so that each difference between the decree and the GL fixtures (tests/golden/
 * render_gl_*.npz) can be attributed, and so that the candidates are in place should a render of the reference show that
 * one of them is what it does.  The HIP path implements none of them.
 *   Clipping: in eye space against z = near, attributes (colour as float, 1/Z) interpolated linearly along the clipped edge
 *   as GL does in clip space; the polygon is fanned from its first vertex.
 *   Multisampling: coverage and depth per sample, colour once per pixel at the pixel centre (GL's default, no centroid),
 *   written to the covered samples that pass the depth test; resolve = mean of the four samples, uncovered ones holding the
 *   clear colour (= the key colour).  `pattern` 0: the Direct3D / Vulkan standard 4x positions (6,2) (14,6) (2,10) (10,14)/16
 *   that desktop GPUs use; 1: SwiftShader's (3,10) (10,13) (13,6) (6,3)/16 in image space (y down), as measured on the pinned GL.
 *   `resolve` 0: (sum + 2) >> 2; 1: SwiftShader's avg(avg(s0,s1), avg(s2,s3)) with avg = (a + b + 1) >> 1.
 *   Ambiguity plane: bit 0 where a fragment that did not win lies within depth_tie_tol (in units of 1/Z) of the winner and
 *   carries another colour -- the pixels whose outcome a GL decides by its depth buffer's resolution; bit 1 where the winning
 *   fragment's triangle spans more than a factor 2 in 1/Z (a rubber-sheet triangle across a depth edge: the perspective-
 *   correct colour there amplifies every rounding of the interpolation, and GLs interpolate differently). */
typedef struct { double X, Y, Z; float c[3]; } orc_glv;     /* eye space + colour */

typedef struct {
    int W, H, ns;
    int ox[4], oy[4];           /* sample offsets inside the pixel, sub-pixel units */
    float* zbuf;                /* [ns][H*W] interpolated 1/Z, 0 = empty */
    uint8_t* rgb;               /* [ns][H*W][3] */
    uint8_t* covered;           /* [ns][H*W] */
    uint8_t* ambiguous;         /* [H*W] or NULL: bit 0 depth near-tie, bit 1 the winner's triangle is steep in 1/Z */
    uint8_t* steep;             /* [ns][H*W] work plane */
    int pass2;                  /* 1: fill `ambiguous` only */
    float tol;
    int cull;
} orc_gl_target;

static void orc_gl_raster(orc_gl_target* t, const float u[3], const float v[3], const float izv[3], const float col[3][3])
{
    const int64_t X0 = orc_snap(u[0]), Y0 = orc_snap(v[0]);
    const int64_t X1 = orc_snap(u[1]), Y1 = orc_snap(v[1]);
    const int64_t X2 = orc_snap(u[2]), Y2 = orc_snap(v[2]);
    int64_t area2 = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
    if (area2 == 0) return;
    if ((t->cull == 1 && area2 > 0) || (t->cull == 2 && area2 < 0)) return;
    const int64_t s = area2 > 0 ? 1 : -1;
    area2 *= s;
    int64_t minX = X0 < X1 ? X0 : X1; if (X2 < minX) minX = X2;
    int64_t maxX = X0 > X1 ? X0 : X1; if (X2 > maxX) maxX = X2;
    int64_t minY = Y0 < Y1 ? Y0 : Y1; if (Y2 < minY) minY = Y2;
    int64_t maxY = Y0 > Y1 ? Y0 : Y1; if (Y2 > maxY) maxY = Y2;
    int64_t px0 = orc_floordiv(minX, ORC_SUBPIX) - 1, px1 = orc_floordiv(maxX, ORC_SUBPIX) + 1;
    int64_t py0 = orc_floordiv(minY, ORC_SUBPIX) - 1, py1 = orc_floordiv(maxY, ORC_SUBPIX) + 1;
    if (px0 < 0) px0 = 0;
    if (py0 < 0) py0 = 0;
    if (px1 > t->W - 1) px1 = t->W - 1;
    if (py1 > t->H - 1) py1 = t->H - 1;
    const int64_t dx0 = s * (X2 - X1), dy0 = s * (Y2 - Y1);
    const int64_t dx1 = s * (X0 - X2), dy1 = s * (Y0 - Y2);
    const int64_t dx2 = s * (X1 - X0), dy2 = s * (Y1 - Y0);
    const float ra = 1.0f / (float)area2;
    const size_t n = (size_t)t->W * t->H;
    const int64_t half = ORC_SUBPIX / 2;
    float izmin = izv[0] < izv[1] ? izv[0] : izv[1], izmax = izv[0] > izv[1] ? izv[0] : izv[1];
    if (izv[2] < izmin) izmin = izv[2];
    if (izv[2] > izmax) izmax = izv[2];
    const uint8_t steep = izmax > 2.0f * izmin;
    for (int64_t py = py0; py <= py1; ++py)
        for (int64_t px = px0; px <= px1; ++px) {
            int hit[4], any = 0;
            float izs[4];
            for (int k = 0; k < t->ns; ++k) {
                const int64_t Xs = px * ORC_SUBPIX + t->ox[k], Ys = py * ORC_SUBPIX + t->oy[k];
                const int64_t w0 = s * ((X2 - X1) * (Ys - Y1) - (Y2 - Y1) * (Xs - X1));
                const int64_t w1 = s * ((X0 - X2) * (Ys - Y2) - (Y0 - Y2) * (Xs - X2));
                const int64_t w2 = s * ((X1 - X0) * (Ys - Y0) - (Y1 - Y0) * (Xs - X0));
                hit[k] = orc_edge_in(w0, dx0, dy0) && orc_edge_in(w1, dx1, dy1) && orc_edge_in(w2, dx2, dy2);
                if (!hit[k]) continue;
                any = 1;
                const float q0 = ((float)w0 * ra) * izv[0], q1 = ((float)w1 * ra) * izv[1], q2 = ((float)w2 * ra) * izv[2];
                izs[k] = (q0 + q1) + q2;
            }
            if (!any) continue;
            /* colour: once per pixel, at the pixel centre (== the sample for ns == 1) */
            const int64_t Xc = px * ORC_SUBPIX + half, Yc = py * ORC_SUBPIX + half;
            const int64_t c0 = s * ((X2 - X1) * (Yc - Y1) - (Y2 - Y1) * (Xc - X1));
            const int64_t c1 = s * ((X0 - X2) * (Yc - Y2) - (Y0 - Y2) * (Xc - X2));
            const int64_t c2 = s * ((X1 - X0) * (Yc - Y0) - (Y1 - Y0) * (Xc - X0));
            const float q0 = ((float)c0 * ra) * izv[0], q1 = ((float)c1 * ra) * izv[1], q2 = ((float)c2 * ra) * izv[2];
            const float izc = (q0 + q1) + q2;
            uint8_t frag[3];
            for (int ch = 0; ch < 3; ++ch) {
                float val;
                if (izc > 0.0f) val = rintf(((q0 * col[0][ch] + q1 * col[1][ch]) + q2 * col[2][ch]) * (1.0f / izc));
                else val = rintf((((float)c0 * ra) * col[0][ch] + ((float)c1 * ra) * col[1][ch]) + ((float)c2 * ra) * col[2][ch]);
                if (!(val >= 0.0f)) val = 0.0f;
                if (val > 255.0f) val = 255.0f;
                frag[ch] = (uint8_t)val;
            }
            const size_t o = (size_t)py * t->W + (size_t)px;
            for (int k = 0; k < t->ns; ++k) {
                if (!hit[k]) continue;
                float* zb = t->zbuf + (size_t)k * n; uint8_t* cv = t->covered + (size_t)k * n; uint8_t* rg = t->rgb + 3 * (size_t)k * n;
                if (t->pass2) {
                    if (cv[o] && fabsf(izs[k] - zb[o]) <= t->tol && memcmp(frag, rg + 3 * o, 3) != 0) t->ambiguous[o] |= 1;
                    continue;
                }
                if (cv[o] && !(izs[k] > zb[o])) continue;
                zb[o] = izs[k]; cv[o] = 1; memcpy(rg + 3 * o, frag, 3);
                if (t->steep) t->steep[(size_t)k * n + o] = steep;
            }
        }
}

/* one triangle given in eye space: clip against the near plane (or drop it whole), project with the render camera, raster */
static void orc_gl_tri(orc_gl_target* t, const orc_eye* e, const orc_glv* a, const orc_glv* b, const orc_glv* c,
                       const orc_vert* va, const orc_vert* vb, const orc_vert* vc, int near_clip)
{
    const orc_glv* in[3] = { a, b, c };
    const orc_vert* vv[3] = { va, vb, vc };
    const int ok[3] = { va->ok, vb->ok, vc->ok };
    if (ok[0] && ok[1] && ok[2]) {
        const float u[3] = { va->u, vb->u, vc->u }, v[3] = { va->v, vb->v, vc->v };
        const float iz[3] = { 1.0f / va->z, 1.0f / vb->z, 1.0f / vc->z };
        const float col[3][3] = { { a->c[0], a->c[1], a->c[2] }, { b->c[0], b->c[1], b->c[2] }, { c->c[0], c->c[1], c->c[2] } };
        orc_gl_raster(t, u, v, iz, col);
        return;
    }
    if (!near_clip || !(ok[0] || ok[1] || ok[2])) return;
    /* Sutherland-Hodgman against Z = near */
    float pu[4], pv[4], piz[4], pc[4][3];
    int np = 0;
    const double zn = (double)ORC_NEAR;
    for (int k = 0; k < 3; ++k) {
        const int k1 = (k + 1) % 3;
        const orc_glv* P = in[k]; const orc_glv* Q = in[k1];
        if (ok[k]) {
            pu[np] = vv[k]->u; pv[np] = vv[k]->v; piz[np] = 1.0f / vv[k]->z;
            for (int ch = 0; ch < 3; ++ch) pc[np][ch] = P->c[ch];
            ++np;
        }
        if (ok[k] != ok[k1]) {
            const double tt = (zn - P->Z) / (Q->Z - P->Z);
            const double X = P->X + tt * (Q->X - P->X), Y = P->Y + tt * (Q->Y - P->Y);
            pu[np] = (float)((double)e->fxr * X / zn + (double)e->cxr);
            pv[np] = (float)((double)e->fyr * Y / zn + (double)e->cyr);
            piz[np] = (float)(1.0 / zn);
            for (int ch = 0; ch < 3; ++ch) pc[np][ch] = (float)((double)P->c[ch] + tt * ((double)Q->c[ch] - (double)P->c[ch]));
            ++np;
        }
    }
    for (int k = 1; k + 1 < np; ++k) {
        const float u[3] = { pu[0], pu[k], pu[k + 1] }, v[3] = { pv[0], pv[k], pv[k + 1] }, iz[3] = { piz[0], piz[k], piz[k + 1] };
        float col[3][3];
        for (int ch = 0; ch < 3; ++ch) { col[0][ch] = pc[0][ch]; col[1][ch] = pc[k][ch]; col[2][ch] = pc[k + 1][ch]; }
        orc_gl_raster(t, u, v, iz, col);
    }
}

/* eye-space position of grid vertex (i, j) (f64; only the clipper reads it) */
static void orc_gl_eye_space(const orc_eye* e, int i, int j, float zsrc, orc_glv* o)
{
    const double gx = (double)((float)j * e->sx), gy = (double)((float)i * e->sy), z = (double)zsrc;
    const double xc = (gx - (double)e->cx) * z / (double)e->fx, yc = (gy - (double)e->cy) * z / (double)e->fy;
    if (!e->general) {
        o->X = xc + (double)e->sign * ((double)e->dl / (double)e->fxr); o->Y = yc; o->Z = z;
        return;
    }
    const float* M = e->M;
    o->X = (M[0] * xc + M[1] * yc) + M[2] * z + M[3];
    o->Y = (M[4] * xc + M[5] * yc) + M[6] * z + M[7];
    o->Z = (M[8] * xc + M[9] * yc) + M[10] * z + M[11];
}

static void orc_gl_render_eye(const orc_params* p, const orc_gl_opts* g, int eye, const float* depth, const uint8_t* color,
                              const uint8_t* tri_invalid, const uint8_t* unused, uint8_t* out_rgb, uint8_t* out_mask, uint8_t* out_amb)
{
    const int W = p->W, H = p->H;
    const size_t n = (size_t)W * H;
    orc_eye e;
    orc_eye_setup(p, eye, &e);
    orc_gl_target t;
    memset(&t, 0, sizeof t);
    t.W = W; t.H = H; t.cull = p->cull;
    t.ns = g->samples == 4 ? 4 : 1;
    if (t.ns == 1) { t.ox[0] = t.oy[0] = ORC_SUBPIX / 2; }
    else {
        static const int pat[2][4][2] = { { { 6, 2 }, { 14, 6 }, { 2, 10 }, { 10, 14 } }, { { 3, 10 }, { 10, 13 }, { 13, 6 }, { 6, 3 } } };
        for (int k = 0; k < 4; ++k) { t.ox[k] = pat[g->pattern ? 1 : 0][k][0] * ORC_SUBPIX / 16; t.oy[k] = pat[g->pattern ? 1 : 0][k][1] * ORC_SUBPIX / 16; }
    }
    t.zbuf = (float*)calloc(n * t.ns, sizeof(float));
    t.rgb = (uint8_t*)calloc(n * t.ns, 3);
    t.covered = (uint8_t*)calloc(n * t.ns, 1);
    t.ambiguous = out_amb;
    t.tol = (float)g->depth_tie_tol;
    if (out_amb) { memset(out_amb, 0, n); t.steep = (uint8_t*)calloc(n * t.ns, 1); }

    orc_vert* V = (orc_vert*)malloc(n * sizeof(orc_vert));
    orc_glv* G = (orc_glv*)malloc(n * sizeof(orc_glv));
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            const size_t k = (size_t)i * W + j;
            V[k] = orc_vertex(&e, i, j, depth[k]);
            orc_gl_eye_space(&e, i, j, depth[k], &G[k]);
            for (int ch = 0; ch < 3; ++ch) G[k].c[ch] = (float)color[3 * k + ch];
        }
    for (t.pass2 = 0; t.pass2 < ((out_amb && g->depth_tie_tol > 0.0) ? 2 : 1); ++t.pass2) {
        if (p->mode == ORC_MODE_POINTS) {
            /* the unit square around the snapped vertex, as two triangles at the vertex's own depth (orc_render_eye's rule
             * is what this gives for one sample at the centre) */
            for (size_t k = 0; k < n; ++k) {
                if (!V[k].ok || (unused && unused[k])) continue;
                const float hs = 0.5f;
                const float cu = (float)orc_snap(V[k].u) / (float)ORC_SUBPIX, cv = (float)orc_snap(V[k].v) / (float)ORC_SUBPIX;
                const float iz[3] = { 1.0f / V[k].z, 1.0f / V[k].z, 1.0f / V[k].z };
                const float col[3][3] = { { G[k].c[0], G[k].c[1], G[k].c[2] }, { G[k].c[0], G[k].c[1], G[k].c[2] }, { G[k].c[0], G[k].c[1], G[k].c[2] } };
                const float u1[3] = { cu - hs, cu - hs, cu + hs }, v1[3] = { cv - hs, cv + hs, cv + hs };
                const float u2[3] = { cu - hs, cu + hs, cu + hs }, v2[3] = { cv - hs, cv + hs, cv - hs };
                const int keep = t.cull; t.cull = 0;
                orc_gl_raster(&t, u1, v1, iz, col);
                orc_gl_raster(&t, u2, v2, iz, col);
                t.cull = keep;
            }
        } else {
            const size_t ncell = (size_t)(W - 1) * (H - 1);
            for (int pass = 0; pass < 2; ++pass)
                for (int i = 0; i < H - 1; ++i)
                    for (int j = 0; j < W - 1; ++j) {
                        if (tri_invalid && tri_invalid[(size_t)pass * ncell + (size_t)i * (W - 1) + j]) continue;
                        const size_t i1 = (size_t)i * W + j, i2 = (size_t)(i + 1) * W + j;
                        const size_t i3 = (size_t)(i + 1) * W + j + 1, i4 = (size_t)i * W + j + 1;
                        const size_t v0 = i1, v1 = pass == 0 ? i2 : i3, v2 = pass == 0 ? i3 : i4;
                        orc_gl_tri(&t, &e, &G[v0], &G[v1], &G[v2], &V[v0], &V[v1], &V[v2], g->near_clip);
                    }
        }
    }
    /* resolve + colour-key hole mask (sr:740, 793) */
    for (size_t k = 0; k < n; ++k) {
        uint8_t c[3];
        if (t.ns == 1) {
            if (t.covered[k]) memcpy(c, t.rgb + 3 * k, 3); else memcpy(c, p->key_rgb, 3);
        } else {
            for (int ch = 0; ch < 3; ++ch) {
                int sv[4];
                for (int q = 0; q < 4; ++q) sv[q] = t.covered[(size_t)q * n + k] ? t.rgb[3 * ((size_t)q * n + k) + ch] : p->key_rgb[ch];
                c[ch] = g->resolve ? (uint8_t)((((sv[0] + sv[1] + 1) >> 1) + ((sv[2] + sv[3] + 1) >> 1) + 1) >> 1)
                                   : (uint8_t)((sv[0] + sv[1] + sv[2] + sv[3] + 2) >> 2);
            }
        }
        const int hole = c[0] == p->key_rgb[0] && c[1] == p->key_rgb[1] && c[2] == p->key_rgb[2];
        out_mask[k] = hole ? 255 : 0;
        if (hole) { out_rgb[3 * k] = out_rgb[3 * k + 1] = out_rgb[3 * k + 2] = 0; }
        else memcpy(out_rgb + 3 * k, c, 3);
    }
    if (t.steep) { for (size_t k = 0; k < n * t.ns; ++k) if (t.steep[k]) out_amb[k % n] |= 2; free(t.steep); }
    free(V); free(G); free(t.zbuf); free(t.rgb); free(t.covered);
}

int orc_render_stereo_gl(const orc_params* p, const orc_gl_opts* g, const uint8_t* depth_rgb, const uint8_t* color_rgb,
                         uint8_t* left_rgb, uint8_t* right_rgb, uint8_t* left_mask, uint8_t* right_mask,
                         uint8_t* left_ambiguous, uint8_t* right_ambiguous)
{
    if (!p || !g || p->W < 2 || p->H < 2) return -1;
    if (p->mode != ORC_MODE_POINTS && p->mode != ORC_MODE_MESH) return -1;
    if (g->samples != 0 && g->samples != 1 && g->samples != 4) return -1;
    orc_set_subpix(p);
    if (g->samples == 4 && ORC_SUBPIX < 16) return -1;
    const int W = p->W, H = p->H;
    const size_t n = (size_t)W * H;
    float* depth = (float*)malloc(n * sizeof(float));
    orc_decode_depth(depth_rgb, W, H, p->max_depth, p->depth_scale, depth);
    uint8_t* tri_invalid = NULL; uint8_t* unused = NULL;
    if (p->remove_edges) {
        tri_invalid = (uint8_t*)malloc(2 * (size_t)(W - 1) * (H - 1));
        unused = (uint8_t*)malloc(n);
        orc_edge_filter(depth, W, H, p->K, p->mode == ORC_MODE_MESH, tri_invalid, unused, NULL);
    }
    orc_gl_render_eye(p, g, 0, depth, color_rgb, tri_invalid, unused, left_rgb, left_mask, left_ambiguous);
    orc_gl_render_eye(p, g, 1, depth, color_rgb, tri_invalid, unused, right_rgb, right_mask, right_ambiguous);
    free(tri_invalid); free(unused); free(depth);
    return 0;
}

int orc_render_stereo(const orc_params* p, const uint8_t* depth_rgb, const uint8_t* color_rgb,
                      uint8_t* left_rgb, uint8_t* right_rgb, uint8_t* left_mask, uint8_t* right_mask,
                      float* left_depth, float* right_depth)
{
    return orc_render_stereo_seed(p, depth_rgb, color_rgb, left_rgb, right_rgb, left_mask, right_mask,
                                  left_depth, right_depth, NULL, NULL);
}

/* The edge-point chain on its own (tests): for every vertex k = i*W + j of a decoded, scaled depth map the rounded
 * pixel of both eyes -- px[k][eye][0..1] = (x, y), INT32_MIN twice where the rounded pixel lies outside the frame or the
 * vertex has depth code 0 --, the f64 depth z[k][eye] the painter's order sorts by and, if `normals` (H*W*3, the removed
 * vertex normals of orc_edge_filter) is given, the un-normalised unprojected normal nrm[k][eye][0..2] (sr:733, 849).
 * Any of px / z / nrm may be NULL. */
void orc_edge_point_chain(const orc_params* p, const float* depth, const double* normals, int32_t* px, double* z, double* nrm)
{
    orc_echain e;
    orc_echain_setup(p, &e);
    const int W = p->W, H = p->H;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            const size_t k = (size_t)i * W + j;
            double P[3], Q[3], A[3], qe[2][3], ae[2][3];
            orc_echain_vertex(&e, i, j, depth[k], P);
            Q[0] = P[0] * e.sW; Q[1] = P[1] * e.sH; Q[2] = P[2];
            orc_echain_eyes(&e, Q, qe[0], qe[1]);
            if (normals && nrm) {
                for (int r = 0; r < 3; ++r) A[r] = normals[3 * k + r] + P[r];
                orc_echain_eyes(&e, A, ae[0], ae[1]);
            }
            for (int eye = 0; eye < 2; ++eye) {
                if (px) {
                    int x, y;
                    int32_t* o = px + 4 * k + 2 * eye;
                    if (depth[k] > ORC_NEAR && orc_echain_pixel(&e, qe[eye], &x, &y)) { o[0] = x; o[1] = y; }
                    else { o[0] = o[1] = INT32_MIN; }
                }
                if (z) z[2 * k + eye] = qe[eye][2];
                if (normals && nrm) for (int r = 0; r < 3; ++r) nrm[6 * k + 3 * eye + r] = ae[eye][r] - qe[eye][r];
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* infill_using_normals (sr:155-240)                                                          */
/* ------------------------------------------------------------------------------------------ */

void orc_infill_using_normals(const uint8_t* color, const uint8_t* hole, const float* normal, int W, int H,
                              int max_steps, uint8_t* out)
{
    const size_t n = (size_t)W * H;
    memcpy(out, color, n * 3);
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const size_t k = (size_t)y * W + x;
            if (!hole[k]) continue;
            const float nx = normal[3 * k], ny = normal[3 * k + 1], nz = normal[3 * k + 2];
            /* sr:176-179: dirs = normal[..., :2] (f32); norms = np.linalg.norm -> sqrt(x*x + y*y) in f32 */
            const float len = sqrtf(nx * nx + ny * ny);
            if (!(len > 1e-6f)) continue;                       /* valid = norms > 1e-6 */
            if (nx == 0.0f && ny == 1.0f && nz == 0.0f) continue;   /* sr:182: green-coded normals are skipped */
            const float dx = nx / len, dy = ny / len;
            const float px = (float)x, py = (float)y;
            for (int t = 1; t <= max_steps; ++t) {
                const float sx = px + dx * (float)t, sy = py + dy * (float)t;    /* sr:205 (f32) */
                const float rx = rintf(sx), ry = rintf(sy);
                if (!(rx >= 0.0f && rx < (float)W && ry >= 0.0f && ry < (float)H)) break;   /* left the image: ray dies */
                const int xi = (int)rx, yi = (int)ry;
                if (hole[(size_t)yi * W + xi]) continue;
                /* sr:220-228: prefer t+2, then t+1, then t */
                for (int dt = 2; dt >= 0; --dt) {
                    const float off = (float)(t + dt);
                    const float qx = rintf(px + dx * off), qy = rintf(py + dy * off);
                    if (!(qx >= 0.0f && qx < (float)W && qy >= 0.0f && qy < (float)H)) continue;
                    const size_t s = (size_t)(int)qy * W + (int)qx;
                    if (hole[s]) continue;
                    memcpy(out + 3 * k, color + 3 * s, 3);
                    break;
                }
                break;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* mark_lower_side (infill_common.py:4-49)                                                    */
/* ------------------------------------------------------------------------------------------ */

void orc_mark_lower_side(const uint8_t* img, int W, int H, int max_steps, uint8_t* out)
{
    memset(out, 0, (size_t)W * H * 3);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t* p = img + 3 * ((size_t)y * W + x);
            if (p[0] == 0 && p[1] == 0 && p[2] == 0) continue;             /* ic:7 valid = non-black */
            const float dx0 = ((float)p[0] / 255.0f) * 2.0f - 1.0f;        /* ic:10 (f32) */
            const float dy0 = ((float)p[1] / 255.0f) * 2.0f - 1.0f;
            const float len = sqrtf(dx0 * dx0 + dy0 * dy0);
            if (!(len > 1e-6f)) continue;                                   /* ic:12 */
            const float dx = dx0 / len, dy = dy0 / len;
            const float fx = (float)x, fy = (float)y;
            for (int t = 1; t < max_steps; ++t) {                           /* ic:20 range(1, max_steps) */
                const float rx = rintf(fx + dx * (float)t), ry = rintf(fy + dy * (float)t);
                if (!(rx >= 0.0f && rx < (float)W && ry >= 0.0f && ry < (float)H)) break;
                const uint8_t* q = img + 3 * ((size_t)(int)ry * W + (int)rx);
                if (!(q[0] == 0 && q[1] == 0 && q[2] == 0)) continue;
                const float bx = rintf(fx + dx * (float)(t - 1)), by = rintf(fy + dy * (float)(t - 1));   /* ic:35-39 */
                if (bx >= 0.0f && by >= 0.0f) {
                    uint8_t* o = out + 3 * ((size_t)(int)by * W + (int)bx);
                    o[0] = 0; o[1] = 0; o[2] = 255;
                }
                break;
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* VR180: convert_to_equirectangular (sr:25-86)                                               */
/* ------------------------------------------------------------------------------------------ */

void orc_equirect_tables(int W, int H, double input_fov_deg, float* mx, float* my)
{
    const double pi = 3.141592653589793;
    const double cx = ((double)W - 1.0) / 2.0, cy = ((double)H - 1.0) / 2.0;      /* sr:41-42 */
    const double half = (input_fov_deg / 2.0) * (pi / 180.0);                      /* np.radians, sr:57 */
    const double fx = cx / tan(half), fy = cy / tan(half);                         /* sr:61-62 */
    for (int x = 0; x < W; ++x) {
        const double theta = ((double)x - cx) / cx * (pi / 2.0);                   /* sr:52 (linspace(0,W-1,W)[x] == x) */
        mx[x] = fabs(theta) <= half ? (float)(fx * tan(theta) + cx) : -1.0f;       /* sr:65-74, 78 */
    }
    for (int y = 0; y < H; ++y) {
        const double phi = ((double)y - cy) / cy * (pi / 2.0);                     /* sr:54 */
        my[y] = fabs(phi) <= half ? (float)(fy * tan(phi) + cy) : -1.0f;
    }
}

void orc_remap_linear(const uint8_t* src, int W, int H, const float* map_x, const float* map_y, uint8_t* dst)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t o = (size_t)y * W + x;
            const int sx = (int)lrintf(map_x[o] * 32.0f), sy = (int)lrintf(map_y[o] * 32.0f);   /* cvRound */
            const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
            const int w[4] = { (32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32 };
            int acc[3] = { 0, 0, 0 };
            for (int t = 0; t < 4; ++t) {
                const int tx = ix + (t & 1), ty = iy + (t >> 1);
                if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;             /* BORDER_CONSTANT, value 0 */
                const uint8_t* s = src + 3 * ((size_t)ty * W + tx);
                for (int c = 0; c < 3; ++c) acc[c] += w[t] * s[c];
            }
            for (int c = 0; c < 3; ++c) dst[3 * o + c] = (uint8_t)((acc[c] + (1 << 14)) >> 15);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* infill-mask completion (sr:803-808, 114-153): TELEA-type inpaint + masked blur             */
/* ------------------------------------------------------------------------------------------ */

/* cv2.getGaussianKernel(6, 0) as published (sigma = 0.3*((n-1)*0.5 - 1) + 0.8 = 1.25, exp in f64, scaled by
 * the reciprocal of the sum), outer product in f64 (g1d @ g1d.T, sr:124-125), rounded to f32 (what
 * cv2.filter2D does with a kernel for an f32 image). */
void orc_masked_blur_kernel(float K[36])
{
    double g[6], sum = 0.0;
    const double sigma = 0.3 * ((6 - 1) * 0.5 - 1.0) + 0.8, scale2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < 6; ++i) { const double x = (double)i - (6 - 1) * 0.5; g[i] = exp(scale2 * x * x); sum += g[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < 6; ++i) g[i] *= sum;
    for (int y = 0; y < 6; ++y) for (int x = 0; x < 6; ++x) K[6 * y + x] = (float)(g[y] * g[x]);
}

void orc_masked_blur(const uint8_t* img, int W, int H, uint8_t* out)
{
    float K[36];
    orc_masked_blur_kernel(K);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float acc[3] = { 0.0f, 0.0f, 0.0f }, wsum = 0.0f;
            for (int ky = 0; ky < 6; ++ky)
                for (int kx = 0; kx < 6; ++kx) {                  /* correlation, anchor (3,3), zero border */
                    const int sx = x + kx - 3, sy = y + ky - 3;
                    if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                    const uint8_t* s = img + 3 * ((size_t)sy * W + sx);
                    const float k = K[6 * ky + kx];
                    for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)s[c];
                    if (s[0] | s[1] | s[2]) wsum = wsum + k;       /* valid_mask = 1 where not black (sr:129-132) */
                }
            const uint8_t* p = img + 3 * ((size_t)y * W + x);
            uint8_t* o = out + 3 * ((size_t)y * W + x);
            const int black = !(p[0] | p[1] | p[2]);
            for (int c = 0; c < 3; ++c) {
                float v = (wsum == 0.0f || black) ? 0.0f : acc[c] / wsum;      /* sr:146-150 */
                if (v < 0.0f) v = 0.0f;
                if (v > 255.0f) v = 255.0f;
                o[c] = (uint8_t)v;                                             /* np.clip(...).astype(np.uint8) */
            }
        }
}

/* Telea's inpainting weights (A. Telea, "An image inpainting technique based on the fast marching method", 2004;
 * the arithmetic follows OpenCV's published icvTeleaInpaintFMM / FastMarching_solve) applied LEVEL-SYNCHRONOUSLY:
 * round r fills every unknown pixel that has a 4-neighbour known at the start of the round (known = original or
 * filled in a round < r), reading only that state.  OpenCV pops one pixel at a time from a heap ordered by T, so
 * its fill order -- and with it the low-order bits of the result -- differs: PARITY UNPINNED, by construction not
 * bit-identical to cv2.inpaint.  Further decrees: T = 0 at every originally known pixel; out-of-image neighbours
 * count as unknown.  stamp: 0 = known from the start, 0xFFFF = unknown, r = filled in round r. */
#define ORC_T_UNKNOWN 0xFFFFu

typedef struct { int W, H; const uint16_t* stamp; const float* T; const uint8_t* img; } orc_telea_state;

static inline int orc_tk(const orc_telea_state* s, int x, int y, unsigned r)      /* known before round r? */
{
    return x >= 0 && x < s->W && y >= 0 && y < s->H && s->stamp[(size_t)y * s->W + x] < r;
}

static float orc_telea_solve(const orc_telea_state* s, int x1, int y1, int x2, int y2, unsigned r)
{
    const int k1 = orc_tk(s, x1, y1, r), k2 = orc_tk(s, x2, y2, r);
    const double a11 = k1 ? (double)s->T[(size_t)y1 * s->W + x1] : 1.0e6, a22 = k2 ? (double)s->T[(size_t)y2 * s->W + x2] : 1.0e6;
    const double m12 = a11 < a22 ? a11 : a22;
    double sol;
    if (k1) {
        if (k2) sol = fabs(a11 - a22) >= 1.0 ? 1.0 + m12 : (a11 + a22 + sqrt(2.0 - (a11 - a22) * (a11 - a22))) * 0.5;
        else sol = 1.0 + a11;
    } else if (k2) sol = 1.0 + a22;
    else sol = 1.0 + m12;
    return (float)sol;
}

static void orc_telea_pixel(const orc_telea_state* s, int x, int y, unsigned r, int radius, float* Tout, uint8_t* rgb)
{
    const int W = s->W;
    float t = orc_telea_solve(s, x, y - 1, x - 1, y, r);
    float c = orc_telea_solve(s, x, y + 1, x - 1, y, r); if (c < t) t = c;
    c = orc_telea_solve(s, x, y - 1, x + 1, y, r); if (c < t) t = c;
    c = orc_telea_solve(s, x, y + 1, x + 1, y, r); if (c < t) t = c;
    *Tout = t;
#define ORC_TT(xx, yy) (s->T[(size_t)(yy) * W + (xx)])
    float gtx, gty;
    if (orc_tk(s, x + 1, y, r)) gtx = orc_tk(s, x - 1, y, r) ? (ORC_TT(x + 1, y) - ORC_TT(x - 1, y)) * 0.5f : ORC_TT(x + 1, y) - t;
    else gtx = orc_tk(s, x - 1, y, r) ? t - ORC_TT(x - 1, y) : 0.0f;
    if (orc_tk(s, x, y + 1, r)) gty = orc_tk(s, x, y - 1, r) ? (ORC_TT(x, y + 1) - ORC_TT(x, y - 1)) * 0.5f : ORC_TT(x, y + 1) - t;
    else gty = orc_tk(s, x, y - 1, r) ? t - ORC_TT(x, y - 1) : 0.0f;
    float Ia[3] = { 0, 0, 0 }, Jx[3] = { 0, 0, 0 }, Jy[3] = { 0, 0, 0 }, sw = 1.0e-20f;
    for (int k = y - radius; k <= y + radius; ++k)
        for (int l = x - radius; l <= x + radius; ++l) {
            if (!orc_tk(s, l, k, r)) continue;
            if ((l - x) * (l - x) + (k - y) * (k - y) > radius * radius) continue;
            const float ry = (float)(y - k), rx = (float)(x - l);
            const float vl = rx * rx + ry * ry;
            const float dst = (float)(1.0 / ((double)vl * sqrt((double)vl)));
            const float lev = (float)(1.0 / (1.0 + fabs((double)(ORC_TT(l, k) - t))));
            float dir = rx * gtx + ry * gty;
            if (fabsf(dir) <= 0.01f) dir = 0.000001f;
            const float w = fabsf((dst * lev) * dir);
            const int xp = orc_tk(s, l + 1, k, r), xm = orc_tk(s, l - 1, k, r), yp = orc_tk(s, l, k + 1, r), ym = orc_tk(s, l, k - 1, r);
            const uint8_t* I0 = s->img + 3 * ((size_t)k * W + l);
            for (int ch = 0; ch < 3; ++ch) {
                float gix, giy;
                if (xp) gix = xm ? (float)((int)I0[3 + ch] - (int)I0[-3 + ch]) * 2.0f : (float)((int)I0[3 + ch] - (int)I0[ch]);
                else gix = xm ? (float)((int)I0[ch] - (int)I0[-3 + ch]) : 0.0f;
                if (yp) giy = ym ? (float)((int)I0[3 * W + ch] - (int)I0[-3 * W + ch]) * 2.0f : (float)((int)I0[3 * W + ch] - (int)I0[ch]);
                else giy = ym ? (float)((int)I0[ch] - (int)I0[-3 * W + ch]) : 0.0f;
                Ia[ch] = Ia[ch] + w * (float)I0[ch];
                Jx[ch] = Jx[ch] - w * (gix * rx);
                Jy[ch] = Jy[ch] - w * (giy * ry);
            }
            sw = sw + w;
        }
#undef ORC_TT
    for (int ch = 0; ch < 3; ++ch) {
        const float jj = Jx[ch] * Jx[ch] + Jy[ch] * Jy[ch];
        const float sat = (float)(((double)(Ia[ch] / sw) + (double)(Jx[ch] + Jy[ch]) / (sqrt((double)jj) + (double)1.0e-20f)) + (double)0.5f);
        float v = rintf(sat);                                   /* saturate_cast<uchar>: cvRound, then clamp */
        if (!(v >= 0.0f)) v = 0.0f;
        if (v > 255.0f) v = 255.0f;
        rgb[ch] = (uint8_t)v;
    }
}

int orc_telea_levels(const uint8_t* img, const uint8_t* mask, const uint8_t* must_fill, int W, int H, int radius,
                     int max_rounds, uint8_t* out)
{
    const size_t n = (size_t)W * H;
    uint16_t* stamp = (uint16_t*)malloc(n * sizeof(uint16_t));
    float* T = (float*)calloc(n, sizeof(float));
    uint32_t* front = (uint32_t*)malloc(n * sizeof(uint32_t));
    float* ft = (float*)malloc(n * sizeof(float));
    uint8_t* frgb = (uint8_t*)malloc(n * 3);
    memcpy(out, img, n * 3);
    long remaining = 0;
    for (size_t k = 0; k < n; ++k) {
        stamp[k] = mask[k] ? ORC_T_UNKNOWN : 0;
        if (mask[k] && (!must_fill || must_fill[k])) ++remaining;
    }
    if (max_rounds > 65000) max_rounds = 65000;
    orc_telea_state s = { W, H, stamp, T, out };
    for (unsigned r = 1; r <= (unsigned)max_rounds && remaining > 0; ++r) {
        size_t nf = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t k = (size_t)y * W + x;
                if (stamp[k] != ORC_T_UNKNOWN) continue;
                if (!(orc_tk(&s, x - 1, y, r) || orc_tk(&s, x + 1, y, r) || orc_tk(&s, x, y - 1, r) || orc_tk(&s, x, y + 1, r))) continue;
                orc_telea_pixel(&s, x, y, r, radius, &ft[nf], frgb + 3 * nf);
                front[nf++] = (uint32_t)k;
            }
        if (nf == 0) break;
        for (size_t q = 0; q < nf; ++q) {                        /* commit after the scan: the round read old state only */
            const size_t k = front[q];
            stamp[k] = (uint16_t)r; T[k] = ft[q]; memcpy(out + 3 * k, frgb + 3 * q, 3);
            if (!must_fill || must_fill[k]) --remaining;
        }
    }
    free(stamp); free(T); free(front); free(ft); free(frgb);
    return (int)remaining;
}

/* sr:803-808 for one eye, from the seed image of orc_render_stereo_seed to the bytes written to the infill-mask
 * video (sr:815-816 / 921-928), including NumPy's float round trips:
 *   lut1[v] = uint8(float64(float32(v)/float32(255)) * 255)      sr:807 into the float64 image, then sr:808's *255
 *   lut2[v] = uint8((float32(v)/float32(255)) * float32(255))    sr:808 .astype('float32')/255.0, then sr:816
 * blur_u8 (optional) receives masked_blur's own uint8 output (the normals --do_basic_infill marches along). */
int orc_finish_infill_mask(const uint8_t* seed, int W, int H, const uint8_t* key_rgb, int max_rounds, uint8_t* out, uint8_t* blur_u8)
{
    const size_t n = (size_t)W * H;
    uint8_t* mask = (uint8_t*)calloc(n + 1, 1); uint8_t* green = (uint8_t*)calloc(n + 1, 1);
    uint8_t* filled = (uint8_t*)malloc(n * 3); uint8_t* merged = (uint8_t*)malloc(n * 3); uint8_t* blur = (uint8_t*)malloc(n * 3);
    for (size_t k = 0; k < n; ++k) {
        const uint8_t* p = seed + 3 * k;
        green[k] = p[0] == key_rgb[0] && p[1] == key_rgb[1] && p[2] == key_rgb[2];      /* sr:803 */
        mask[k] = green[k] || !(p[0] | p[1] | p[2]);                                     /* sr:804-805 */
    }
    const int remaining = orc_telea_levels(seed, mask, green, W, H, 3, max_rounds, filled);   /* sr:806 */
    for (size_t k = 0; k < n; ++k)
        for (int c = 0; c < 3; ++c) {
            const uint8_t v = filled[3 * k + c];
            merged[3 * k + c] = green[k] ? (uint8_t)((double)((float)v / 255.0f) * 255.0) : seed[3 * k + c];   /* sr:807-808 */
        }
    orc_masked_blur(merged, W, H, blur);
    if (blur_u8) memcpy(blur_u8, blur, n * 3);
    for (size_t k = 0; k < 3 * n; ++k) out[k] = (uint8_t)(((float)blur[k] / 255.0f) * 255.0f);               /* sr:808, 816 */
    free(mask); free(green); free(filled); free(merged); free(blur);
    return remaining;
}

/* The SEQUENTIAL fast-marching order of cv2.inpaint(INPAINT_TELEA), with the same per-pixel estimator and the same
 * decrees as orc_telea_levels (T = 0 at known pixels, out-of-image = unknown): a heap ordered by T (ties: first in,
 * first out), one pixel popped at a time, its unknown 4-neighbours estimated from everything not INSIDE at that
 * moment.  Not used for parity -- it exists to MEASURE how far the level-synchronous order of orc_telea_levels /
 * the device moves the result away from the sequential one (tests/test_oracle_golden.py reports the statistics). */
typedef struct { float t; uint32_t seq; uint32_t idx; } orc_heap_item;

static void orc_heap_push(orc_heap_item* h, size_t* n, orc_heap_item it)
{
    size_t k = (*n)++;
    h[k] = it;
    while (k > 0) {
        const size_t p = (k - 1) / 2;
        if (h[p].t < h[k].t || (h[p].t == h[k].t && h[p].seq < h[k].seq)) break;
        const orc_heap_item tmp = h[p]; h[p] = h[k]; h[k] = tmp; k = p;
    }
}

static orc_heap_item orc_heap_pop(orc_heap_item* h, size_t* n)
{
    const orc_heap_item top = h[0];
    h[0] = h[--(*n)];
    size_t k = 0;
    for (;;) {
        size_t l = 2 * k + 1, r = l + 1, m = k;
        if (l < *n && (h[l].t < h[m].t || (h[l].t == h[m].t && h[l].seq < h[m].seq))) m = l;
        if (r < *n && (h[r].t < h[m].t || (h[r].t == h[m].t && h[r].seq < h[m].seq))) m = r;
        if (m == k) break;
        const orc_heap_item tmp = h[m]; h[m] = h[k]; h[k] = tmp; k = m;
    }
    return top;
}

void orc_telea_fmm(const uint8_t* img, const uint8_t* mask, int W, int H, int radius, uint8_t* out)
{
    const size_t n = (size_t)W * H;
    /* stamp doubles as the flag: 0 = not INSIDE (known or band), 0xFFFF = INSIDE; orc_telea_pixel() with r = 1 then
     * reads exactly "everything not INSIDE right now" */
    uint16_t* stamp = (uint16_t*)malloc(n * sizeof(uint16_t));
    float* T = (float*)calloc(n, sizeof(float));
    orc_heap_item* heap = (orc_heap_item*)malloc((4 * n + 4) * sizeof(orc_heap_item));
    size_t hn = 0;
    uint32_t seq = 0;
    memcpy(out, img, n * 3);
    for (size_t k = 0; k < n; ++k) stamp[k] = mask[k] ? ORC_T_UNKNOWN : 0;
    orc_telea_state s = { W, H, stamp, T, out };
    const int dx[4] = { 0, -1, 0, 1 }, dy[4] = { -1, 0, 1, 0 };        /* OpenCV's neighbour order: (i-1,j), (i,j-1), (i+1,j), (i,j+1) */
    for (int y = 0; y < H; ++y)                                          /* the initial band: known pixels next to the mask, T = 0 */
        for (int x = 0; x < W; ++x) {
            if (stamp[(size_t)y * W + x]) continue;
            int band = 0;
            for (int d = 0; d < 4; ++d) {
                const int xx = x + dx[d], yy = y + dy[d];
                if (xx >= 0 && xx < W && yy >= 0 && yy < H && stamp[(size_t)yy * W + xx]) band = 1;
            }
            if (band) { const orc_heap_item it = { 0.0f, seq++, (uint32_t)((size_t)y * W + x) }; orc_heap_push(heap, &hn, it); }
        }
    while (hn) {
        const orc_heap_item it = orc_heap_pop(heap, &hn);
        const int y = (int)(it.idx / (uint32_t)W), x = (int)(it.idx % (uint32_t)W);
        for (int d = 0; d < 4; ++d) {
            const int xx = x + dx[d], yy = y + dy[d];
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const size_t q = (size_t)yy * W + xx;
            if (stamp[q] != ORC_T_UNKNOWN) continue;
            float t; uint8_t rgb[3];
            orc_telea_pixel(&s, xx, yy, 1, radius, &t, rgb);
            T[q] = t; memcpy(out + 3 * q, rgb, 3); stamp[q] = 0;          /* BAND: from now on a source for others */
            const orc_heap_item nq = { t, seq++, (uint32_t)q };
            orc_heap_push(heap, &hn, nq);
        }
    }
    free(stamp); free(T); free(heap);
}

/* BAND-synchronous fast marching (r05 experiment, verdict r04 item 5a): the sequential march of orc_telea_fmm with its pops
 * grouped by arrival time.  Step k pops -- together -- every BAND pixel with T < k * delta, then estimates -- together, from the
 * state at that moment -- every INSIDE pixel that has one of the popped pixels as a 4-neighbour.  delta -> 0 is the heap order
 * (up to ties); delta = 1 is close to the level-synchronous rounds of orc_telea_levels (T grows by about one per ring), except
 * that diagonal directions, where T grows by ~0.7 per 4-connected ring, run ahead as they do under the heap.  A step is what a
 * level is on the device: two dependent launches, so 1 / delta is also the factor on the completion's launch-bound time.
 * Returns the number of steps that estimated at least one pixel.  Not used for parity: tests/report_infill_order_downstream.py. */
int orc_telea_bands(const uint8_t* img, const uint8_t* mask, int W, int H, int radius, float delta, uint8_t* out)
{
    const size_t n = (size_t)W * H;
    uint16_t* stamp = (uint16_t*)malloc(n * sizeof(uint16_t));          /* 0 = not INSIDE (known or band), 0xFFFF = INSIDE */
    uint8_t* popped = (uint8_t*)calloc(n, 1);                            /* BAND pixel already popped (KNOWN) */
    float* T = (float*)calloc(n, sizeof(float));
    uint32_t* band = (uint32_t*)malloc(n * sizeof(uint32_t));            /* BAND pixels not yet popped */
    uint32_t* fresh = (uint32_t*)malloc(n * sizeof(uint32_t));
    uint32_t* cand = (uint32_t*)malloc(n * sizeof(uint32_t));
    uint8_t* seen = (uint8_t*)calloc(n, 1);
    float* ft = (float*)malloc(n * sizeof(float));
    uint8_t* frgb = (uint8_t*)malloc(n * 3);
    size_t nb = 0;
    int steps = 0;
    memcpy(out, img, n * 3);
    for (size_t k = 0; k < n; ++k) stamp[k] = mask[k] ? ORC_T_UNKNOWN : 0;
    orc_telea_state s = { W, H, stamp, T, out };
    const int dx[4] = { 0, -1, 0, 1 }, dy[4] = { -1, 0, 1, 0 };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (stamp[(size_t)y * W + x]) continue;
            int b = 0;
            for (int d = 0; d < 4; ++d) {
                const int xx = x + dx[d], yy = y + dy[d];
                if (xx >= 0 && xx < W && yy >= 0 && yy < H && stamp[(size_t)yy * W + xx]) b = 1;
            }
            if (b) band[nb++] = (uint32_t)((size_t)y * W + x);
        }
    double theta = 0.0;
    while (nb) {
        float tmin = 1.0e30f;
        for (size_t q = 0; q < nb; ++q) if (T[band[q]] < tmin) tmin = T[band[q]];
        theta += (double)delta;
        if ((double)tmin >= theta) theta = (floor((double)tmin / (double)delta) + 1.0) * (double)delta;      /* skip empty steps */
        size_t nf = 0, keep = 0, nc = 0;
        for (size_t q = 0; q < nb; ++q) {
            if ((double)T[band[q]] < theta) { fresh[nf++] = band[q]; popped[band[q]] = 1; }
            else band[keep++] = band[q];
        }
        nb = keep;
        for (size_t q = 0; q < nf; ++q) {
            const int y = (int)(fresh[q] / (uint32_t)W), x = (int)(fresh[q] % (uint32_t)W);
            for (int d = 0; d < 4; ++d) {
                const int xx = x + dx[d], yy = y + dy[d];
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                const size_t k = (size_t)yy * W + xx;
                if (stamp[k] != ORC_T_UNKNOWN || seen[k]) continue;
                seen[k] = 1; cand[nc++] = (uint32_t)k;
            }
        }
        for (size_t q = 0; q < nc; ++q)
            orc_telea_pixel(&s, (int)(cand[q] % (uint32_t)W), (int)(cand[q] / (uint32_t)W), 1, radius, &ft[q], frgb + 3 * q);
        for (size_t q = 0; q < nc; ++q) {                                /* commit after the scan: the step read old state only */
            const size_t k = cand[q];
            stamp[k] = 0; T[k] = ft[q]; memcpy(out + 3 * k, frgb + 3 * q, 3);
            band[nb++] = (uint32_t)k;
        }
        if (nc) ++steps;
    }
    free(stamp); free(popped); free(T); free(band); free(fresh); free(cand); free(seen); free(ft); free(frgb);
    return steps;
}

/* The heap order of orc_telea_fmm as a CONSERVATIVE PARALLEL SIMULATION (r06, verdict r05 item 3): the same result, bit for bit,
 * from steps whose parts are order-free, and the numbers that say what a device version of it would cost.
 *
 * Look-ahead.  A pixel q is estimated once, when the first of its 4-neighbours is popped (its "parent" p, the heap's minimum at that
 * moment).  Every other non-INSIDE 4-neighbour of q is then either still in the band (T >= T(p)) or original (T = 0: then q's
 * parent is an original pixel and T(p) = 0).  FastMarching_solve gives min + 1/sqrt(2) at the least, so T(q) >= T(p) + 0.7071: a
 * pixel activated by a pop of the window [k d, (k+1) d), d = 0.70, is itself popped in a LATER window.  Hence one window at a time:
 *   1. the window's pops = the band pixels with T < (k+1) d, SORTED by (T, S) -- S = the activation order, below;
 *   2. every INSIDE pixel next to a pop is activated; its order key A = 4 rank(parent) + d, parent = the adjacent pop of least rank,
 *      d = OpenCV's neighbour index of q as seen from the parent (up, left, down, right): per pixel, order-free;
 *   3. T and colour of the activated pixels, each reading the pixels activated BEFORE it (earlier windows: all; this window:
 *      smaller A): a dependency graph whose edges join pixels at most 4 apart (the radius-3 disc and the 4-neighbours its image
 *      gradients read).  Here it is walked in A order; a device walks it by readiness.  S = window << 32 | A.
 * stats[0] windows, [1] most pops in a window, [2] all pops, [3] sum over the windows of the longest T-dependency chain inside the
 * window (4-neighbours), [4] the same for the colour estimate (distance <= 4), [5] longest single-window colour chain, [6] pixels
 * whose T fell inside their own window (must be 0: the look-ahead), [7] most activations in a window. */
typedef struct { float t; uint64_t s; uint32_t idx; } orc_win_item;
static int orc_win_cmp(const void* a, const void* b)
{
    const orc_win_item* x = (const orc_win_item*)a; const orc_win_item* y = (const orc_win_item*)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->s < y->s ? -1 : (x->s > y->s ? 1 : 0);
}
typedef struct { uint32_t a; uint32_t idx; } orc_act_item;
static int orc_act_cmp(const void* a, const void* b)
{
    const orc_act_item* x = (const orc_act_item*)a; const orc_act_item* y = (const orc_act_item*)b;
    return x->a < y->a ? -1 : (x->a > y->a ? 1 : 0);
}

void orc_telea_windows(const uint8_t* img, const uint8_t* mask, int W, int H, int radius, uint8_t* out, float* T_out, uint64_t stats[8])
{
    const size_t n = (size_t)W * H;
    const double delta = 0.70;
    uint16_t* stamp = (uint16_t*)malloc(n * sizeof(uint16_t));           /* 0 = not INSIDE, 0xFFFF = INSIDE */
    float* T = (float*)calloc(n, sizeof(float));
    uint64_t* S = (uint64_t*)calloc(n, sizeof(uint64_t));
    uint32_t* akey = (uint32_t*)malloc(n * sizeof(uint32_t));            /* activation key in the current window, ~0 = none */
    uint32_t* awin = (uint32_t*)calloc(n, sizeof(uint32_t));             /* window (1-based) in which the pixel was activated */
    uint32_t* dT = (uint32_t*)calloc(n, sizeof(uint32_t));               /* chain depths inside the activation window */
    uint32_t* dC = (uint32_t*)calloc(n, sizeof(uint32_t));
    uint32_t* band = (uint32_t*)malloc(n * sizeof(uint32_t));
    orc_win_item* pops = (orc_win_item*)malloc(n * sizeof(orc_win_item));
    orc_act_item* acts = (orc_act_item*)malloc(n * sizeof(orc_act_item));
    size_t nb = 0;
    memset(stats, 0, 8 * sizeof(uint64_t));
    memset(akey, 0xFF, n * sizeof(uint32_t));
    memcpy(out, img, n * 3);
    for (size_t k = 0; k < n; ++k) stamp[k] = mask[k] ? ORC_T_UNKNOWN : 0;
    orc_telea_state st = { W, H, stamp, T, out };
    const int dx[4] = { 0, -1, 0, 1 }, dy[4] = { -1, 0, 1, 0 };
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (stamp[(size_t)y * W + x]) continue;
            int b = 0;
            for (int d = 0; d < 4; ++d) {
                const int xx = x + dx[d], yy = y + dy[d];
                if (xx >= 0 && xx < W && yy >= 0 && yy < H && stamp[(size_t)yy * W + xx]) b = 1;
            }
            if (b) { const size_t k = (size_t)y * W + x; S[k] = (uint64_t)k; band[nb++] = (uint32_t)k; }
        }
    uint32_t win = 0;
    while (nb) {
        float tmin = 1.0e30f;
        for (size_t q = 0; q < nb; ++q) if (T[band[q]] < tmin) tmin = T[band[q]];
        const double hi = (floor((double)tmin / delta) + 1.0) * delta;
        ++win;
        /* 1. the window's pops, sorted */
        size_t np = 0, keep = 0;
        for (size_t q = 0; q < nb; ++q) {
            const uint32_t k = band[q];
            if ((double)T[k] < hi) { pops[np].t = T[k]; pops[np].s = S[k]; pops[np].idx = k; ++np; }
            else band[keep++] = k;
        }
        nb = keep;
        qsort(pops, np, sizeof(orc_win_item), orc_win_cmp);
        /* 2. activation: per INSIDE neighbour the least 4 rank + d (the loop order is irrelevant: a minimum) */
        size_t na = 0;
        for (size_t r = 0; r < np; ++r) {
            const int y = (int)(pops[r].idx / (uint32_t)W), x = (int)(pops[r].idx % (uint32_t)W);
            for (int d = 0; d < 4; ++d) {
                const int xx = x + dx[d], yy = y + dy[d];
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                const size_t k = (size_t)yy * W + xx;
                if (stamp[k] != ORC_T_UNKNOWN) continue;
                const uint32_t a = (uint32_t)(4 * r + (size_t)d);
                if (akey[k] == 0xFFFFFFFFu) acts[na++].idx = (uint32_t)k;
                if (a < akey[k]) akey[k] = a;
            }
        }
        for (size_t q = 0; q < na; ++q) acts[q].a = akey[acts[q].idx];
        qsort(acts, na, sizeof(orc_act_item), orc_act_cmp);
        /* 3. the estimates, in activation order (every read is of a pixel activated before) */
        uint32_t maxT = 0, maxC = 0;
        for (size_t q = 0; q < na; ++q) {
            const size_t k = acts[q].idx;
            const int y = (int)(k / (size_t)W), x = (int)(k % (size_t)W);
            float t; uint8_t rgb[3];
            orc_telea_pixel(&st, x, y, 1, radius, &t, rgb);
            uint32_t depT = 0, depC = 0;
            for (int yy = y - 4; yy <= y + 4; ++yy)
                for (int xx = x - 4; xx <= x + 4; ++xx) {
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    const size_t m = (size_t)yy * W + xx;
                    if (awin[m] != win || stamp[m] != 0) continue;                      /* activated in this window, before q */
                    if ((xx - x) * (xx - x) + (yy - y) * (yy - y) <= 16 && dC[m] > depC) depC = dC[m];
                    if (abs(xx - x) + abs(yy - y) == 1 && dT[m] > depT) depT = dT[m];
                }
            dT[k] = depT + 1; dC[k] = depC + 1; awin[k] = win;
            if (dT[k] > maxT) maxT = dT[k];
            if (dC[k] > maxC) maxC = dC[k];
            T[k] = t; memcpy(out + 3 * k, rgb, 3); stamp[k] = 0;
            S[k] = ((uint64_t)win << 32) | acts[q].a;
            akey[k] = 0xFFFFFFFFu;
            if ((double)t < hi) ++stats[6];
            band[nb++] = (uint32_t)k;
        }
        ++stats[0];
        if (np > stats[1]) stats[1] = np;
        stats[2] += np; stats[3] += maxT; stats[4] += maxC;
        if (maxC > stats[5]) stats[5] = maxC;
        if (na > stats[7]) stats[7] = na;
    }
    if (T_out) memcpy(T_out, T, n * sizeof(float));
    free(stamp); free(T); free(S); free(akey); free(awin); free(dT); free(dC); free(band); free(pops); free(acts);
}

/* ------------------------------------------------------------------------------------------ */
/* basic_nomal_infill.normal_infill (bni:87-119) and the helpers it is made of                 */
/* ------------------------------------------------------------------------------------------ */

static inline int orc_reflect101(int p, int len)        /* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

void orc_box_blur4(const uint8_t* img, int W, int H, uint8_t* out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                int sum = 0;
                for (int dy = -2; dy <= 1; ++dy)
                    for (int dx = -2; dx <= 1; ++dx)
                        sum += img[3 * ((size_t)orc_reflect101(y + dy, H) * W + orc_reflect101(x + dx, W)) + c];
                int q = sum >> 4;                                 /* cvRound(sum * (1/16)): half to even */
                const int rem = sum & 15;
                if (rem > 8 || (rem == 8 && (q & 1))) ++q;
                out[3 * ((size_t)y * W + x) + c] = (uint8_t)q;
            }
}

void orc_dilate_cross(const uint8_t* mask, int W, int H, int iterations, uint8_t* out)
{
    const size_t n = (size_t)W * H;
    uint8_t* cur = (uint8_t*)malloc(n + 1);
    for (size_t k = 0; k < n; ++k) out[k] = mask[k] ? 1 : 0;
    for (int it = 0; it < iterations; ++it) {
        memcpy(cur, out, n);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t k = (size_t)y * W + x;
                if (cur[k]) continue;
                if ((x > 0 && cur[k - 1]) || (x + 1 < W && cur[k + 1]) || (y > 0 && cur[k - W]) || (y + 1 < H && cur[k + W])) out[k] = 1;
            }
    }
    free(cur);
}

void orc_blur_under_mask(const uint8_t* img, const uint8_t* mask, int W, int H, uint8_t* out)
{
    float K[36];
    orc_masked_blur_kernel(K);                                  /* bni:58-59: the same 6 x 6 kernel as masked_blur */
    memcpy(out, img, (size_t)W * H * 3);                        /* bni:82: outside the mask the image stays */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!mask[(size_t)y * W + x]) continue;
            float acc[3] = { 0.0f, 0.0f, 0.0f }, wsum = 0.0f;
            for (int ky = 0; ky < 6; ++ky)
                for (int kx = 0; kx < 6; ++kx) {                /* correlation, anchor (3,3), zero border (bni:68-71) */
                    const int sx = x + kx - 3, sy = y + ky - 3;
                    if (sx < 0 || sx >= W || sy < 0 || sy >= H) continue;
                    if (!mask[(size_t)sy * W + sx]) continue;   /* img_f * m: an unmasked tap adds k * 0 */
                    const uint8_t* s = img + 3 * ((size_t)sy * W + sx);
                    const float k = K[6 * ky + kx];
                    for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k * (float)s[c];
                    wsum = wsum + k;
                }
            uint8_t* o = out + 3 * ((size_t)y * W + x);
            for (int c = 0; c < 3; ++c) {
                float v = acc[c] / wsum;                        /* the pixel itself is masked: wsum > 0 (bni:74-77) */
                if (v < 0.0f) v = 0.0f;
                if (v > 255.0f) v = 255.0f;
                o[c] = (uint8_t)v;                              /* np.clip(...).astype(np.uint8) (bni:85) */
            }
        }
}

void orc_normal_infill(const uint8_t* img, const uint8_t* infill_mask, int W, int H, uint8_t* out, uint8_t* stages)
{
    const size_t n = (size_t)W * H;
    uint8_t* bg = (uint8_t*)malloc(n + 1);
    uint8_t* work = (uint8_t*)malloc(3 * n + 1);
    uint8_t* blur = (uint8_t*)malloc(3 * n + 1);
    uint8_t* filled = (uint8_t*)malloc(3 * n + 1);
    uint8_t* box = (uint8_t*)malloc(3 * n + 1);
    uint8_t* blue = (uint8_t*)malloc(3 * n + 1);
    uint8_t* marks = (uint8_t*)malloc(n + 1);
    uint8_t* grown = (uint8_t*)malloc(n + 1);
    float* normal = (float*)malloc(3 * n * sizeof(float) + 4);
    memcpy(work, img, 3 * n);
    for (size_t k = 0; k < n; ++k) {
        const uint8_t* m = infill_mask + 3 * k;
        bg[k] = (m[0] != 0 && m[1] != 0 && m[2] != 0) ? 1 : 0;                  /* bni:88: np.all(mask != black, axis=-1) */
        if (bg[k]) { work[3 * k] = 0; work[3 * k + 1] = 0; work[3 * k + 2] = 0; }   /* bni:91 */
        for (int c = 0; c < 3; ++c) normal[3 * k + c] = (((float)m[c] / 255.0f) * 2.0f) - 1.0f;   /* bni:94 (f32) */
    }
    orc_masked_blur(work, W, H, blur);                                           /* bni:98 */
    orc_infill_using_normals(blur, bg, normal, W, H, 400, filled);               /* bni:101 */
    orc_box_blur4(filled, W, H, box);                                            /* bni:104 */
    for (size_t k = 0; k < n; ++k) if (bg[k]) memcpy(work + 3 * k, box + 3 * k, 3);   /* bni:107 */
    orc_mark_lower_side(infill_mask, W, H, 30, blue);                            /* bni:111 */
    for (size_t k = 0; k < n; ++k) marks[k] = (blue[3 * k] == 0 && blue[3 * k + 1] == 0 && blue[3 * k + 2] == 255) ? 1 : 0;   /* bni:112 */
    orc_dilate_cross(marks, W, H, 6, grown);                                     /* bni:115 */
    orc_blur_under_mask(work, grown, W, H, out);                                 /* bni:118 */
    if (stages) {
        memcpy(stages, blur, 3 * n); memcpy(stages + 3 * n, filled, 3 * n); memcpy(stages + 6 * n, work, 3 * n);
        memcpy(stages + 9 * n, bg, n); memcpy(stages + 10 * n, grown, n);
    }
    free(bg); free(work); free(blur); free(filled); free(box); free(blue); free(marks); free(grown); free(normal);
}
