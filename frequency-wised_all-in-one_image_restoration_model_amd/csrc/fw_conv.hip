// Convolutional plug-ins of the model seam (K13 of SURVEY.md section 2.2; BASELINE configs[0]):
//   ResNetEncoder / ResBlock        net/encoder_ResNet.py:4-47     3x3 (stride 1 / 2) and 1x1 convolutions, BatchNorm2d, LeakyReLU(0.1)
//   DGRN = ResNetDecoder            net/decoder_DGRN.py:9-158      56 plain 3x3 convolutions 64 -> 64, SFT 1x1 MLPs, DGM combine
//   DCN_layer                       net/utils/deform_conv.py:10-67 offset / mask convolution + modulated deformable convolution (DCNv2)
//
// MI355X design.  Activations are token-major ("NHWC"): row = (image, y, x), channels contiguous, row stride a multiple of 16 bytes.
//   * 3x3 convolutions are IMPLICIT GEMMs on MFMA: a workgroup of 4 waves owns a 4 x 16 patch of output pixels; the input patch with
//     its 1-pixel halo (64 channels at a time) is staged ONCE in LDS, zero-padded at the image border, and every kernel tap reads its
//     MFMA operand from that image at a shifted address (rows of the operand = 16 consecutive output pixels of one row; for stride 2
//     simply every second LDS pixel).  No im2col buffer exists: the activation is read once per 64-channel group.  The weight panel
//     [Cout][9 * Cin] (tap-major) is read as ready-made MFMA fragments from L2 / L1 (73 KB for 64 -> 64, shared by every workgroup).
//     MFMA roles as in fw_gemm: A = weights (rows = output channels), B = pixels, so a lane ends with 4 consecutive output channels
//     of one pixel -> 8 / 16-byte stores.  Epilogue: bias, LeakyReLU, residual add.  The same kernel computes input gradients of
//     stride-1 convolutions (flipped, transposed weight panel).  Channels [0, Cin1) may come from one tensor and [Cin1, Cin) from a
//     second one: the offset convolution of DCN reads cat[x, inter] (deform_conv.py:57) without materialising it.
//   * weight gradients and the input gradient of the four stride-2 convolutions go through an explicit im2col (fw_im2col3) and the
//     split-K GEMMs of fw_gemm.hip.
//   * DCNv2: the bilinear, mask-modulated gather writes the [tokens][9 * Cin] operand of a plain GEMM (fw_dcn_im2col); the backward
//     kernel (one wave per pixel and tap) scatters d(input) with f32 atomics and reduces d(offset), d(mask) over the channels.
//     The reference ends in `assert False` here (mmcv absent): parity unpinned, anchored by known-answer tests.
//   * BatchNorm2d on token-major maps: column statistics by block partials + atomics, then one normalise / affine / residual /
//     LeakyReLU pass; backward in the same two-pass form.
#include "fw_common.h"

namespace {

constexpr int TPB = 256;
FW_DEV long gtid() { return (long)blockIdx.x * blockDim.x + threadIdx.x; }
FW_DEV long gstride() { return (long)gridDim.x * blockDim.x; }
static inline int grid_for(long n, int cap = 262144) { long g = (n + TPB - 1) / TPB; return (int)(g < 1 ? 1 : (g > cap ? cap : g)); }

template <typename T> FW_DEV void load_vec(const T* p, float* f) {          // E16 elements
    unpack16<T>(*reinterpret_cast<const uint4*>(p), f);
}
template <typename T> FW_DEV void store_vec(T* p, const float* f) { *reinterpret_cast<uint4*>(p) = pack16<T>(f); }

// =====================================================================================================
// implicit-GEMM 3x3 convolution
// =====================================================================================================
struct ConvArgs {
    const char* x; long ldx;            // channels [0, cin1)
    const char* x2; long ldx2;          // channels [cin1, cin) (or null)
    int cin, cin1;                      // multiples of the 64-byte chunk
    const char* w;                      // T [coutp][9 * cin], tap-major; coutp = roundup(cout, 16)
    const float* bias;                  // [cout] or null
    char* out; long ldo; int out_f32;
    const char* res; long ldr;          // T residual [Mo][ldr] or null
    int cout;
    int B, H, W, Ho, Wo, stride;
    int taps;                           // bit k set: tap k = ky * 3 + kx is computed (0x1ff = 3x3, 0x010 = a 1x1 convolution)
    int act; float slope;
};

constexpr int CT_Y = 4, CT_X = 16;      // output pixels per workgroup: 4 rows (one per wave) x 16 columns

template <typename T, int S>
__global__ __launch_bounds__(256) void conv3x3_kernel(ConvArgs a) {
    constexpr int SZ = TT<T>::SZ, E = TT<T>::E16;
    constexpr int GC = 64;                               // channels per staged group
    constexpr int KC = GC * SZ / 64;                     // 64-byte chunks per group
    constexpr int LDR = GC * SZ + 16;                    // bytes per LDS pixel
    constexpr int PY = (CT_Y - 1) * S + 3, PX = (CT_X - 1) * S + 3;
    constexpr int GR = GC / E;                           // 16-byte granules per pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = lane_id(), wv = threadIdx.x >> 6;
    const int tx = (a.Wo + CT_X - 1) / CT_X, ty = (a.Ho + CT_Y - 1) / CT_Y;
    const long ntile = (long)a.B * ty * tx;
    const int ncg = (a.cout + 63) / 64;                  // output-channel tiles of 64
    const int kw = 9 * a.cin;                            // row length of the weight panel (elements)
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int b = (int)(tile / (ty * tx)), r = (int)(tile % (ty * tx));
        const int oy0 = (r / tx) * CT_Y, ox0 = (r % tx) * CT_X;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        for (int cg = 0; cg < ncg; ++cg) {
            const int co0 = cg * 64;
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ci0 = 0; ci0 < a.cin; ci0 += GC) {
                // ---- stage the halo patch of channels ci0 .. ci0+63 (two-phase: all loads from clamped coordinates, then selects)
                const bool second = a.x2 != nullptr && ci0 >= a.cin1;
                const char* src = second ? a.x2 : a.x;
                const long ld = second ? a.ldx2 : a.ldx;
                const int cbase = second ? ci0 - a.cin1 : ci0;
                const int gcv = min(GC, a.cin - ci0) / E;            // valid granules of this group
                __syncthreads();                                      // the previous group's fragments are consumed
                for (int idx0 = 0; idx0 < PY * PX * GR; idx0 += 256 * 4) {
                    uint4 v[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = idx0 + u * 256 + threadIdx.x;
                        const int pix = idx / GR, g = idx % GR;
                        const int py = pix / PX, px = pix % PX;
                        const int iy = iy0 + py, ix = ix0 + px;
                        ok[u] = idx < PY * PX * GR && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && g < gcv;
                        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
                        const long row = ((long)b * a.H + cy) * a.W + cx;
                        v[u] = *reinterpret_cast<const uint4*>(src + (row * ld + cbase + (g < gcv ? g : 0) * E) * SZ);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = idx0 + u * 256 + threadIdx.x;
                        if (idx < PY * PX * GR)
                            *reinterpret_cast<uint4*>(smem + (idx / GR) * LDR + (idx % GR) * 16) = ok[u] ? v[u] : make_uint4(0, 0, 0, 0);
                    }
                }
                __syncthreads();
                // ---- 9 taps x KC chunks: B = 16 output pixels of row wv at the tap's shift, A = weight rows from global memory
#pragma unroll 1
                for (int tap = 0; tap < 9; ++tap) {
                    if (!((a.taps >> tap) & 1)) continue;
                    const int ky = tap / 3, kx = tap % 3;
                    const char* brow = smem + ((wv * S + ky) * PX + kx) * LDR;
                    const T* wt = reinterpret_cast<const T*>(a.w) + (long)tap * a.cin + ci0 + ((l >> 4) * E);
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        if (ci0 + c * (64 / SZ) >= a.cin) break;
                        const uint4 bf = frag_kc(brow, S * LDR, 0, c);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const int co = co0 + nt * 16 + (l & 15);
                            if (co0 + nt * 16 < a.cout) {                                  // wave-uniform
                                const uint4 af = *reinterpret_cast<const uint4*>(wt + (long)co * kw + c * (64 / SZ));
                                mma_chunk<T>(acc[nt], af, bf);
                            }
                        }
                    }
                }
            }
            // ---- epilogue: lane holds 4 consecutive output channels of pixel (oy0 + wv, ox0 + (l & 15))
            const int oy = oy0 + wv, ox = ox0 + (l & 15);
            if (oy < a.Ho && ox < a.Wo) {
                const long orow = ((long)b * a.Ho + oy) * a.Wo + ox;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int c0 = co0 + nt * 16 + ((l >> 4) << 2);
                    if (c0 >= a.cout) continue;
                    const bool full = c0 + 4 <= a.cout;                                // wave-uniform per nt except at a ragged tail
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q] = acc[nt][q] + ((a.bias && c0 + q < a.cout) ? a.bias[c0 + q] : 0.f);
                        if (a.act == 1) v[q] = lrelu_f(v[q], a.slope);
                    }
                    if (a.res) {
                        const T* rp = reinterpret_cast<const T*>(a.res) + orow * a.ldr + c0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (c0 + q < a.cout) v[q] += TT<T>::ld(rp + q);
                    }
                    if (a.out_f32) {
                        float* op = reinterpret_cast<float*>(a.out) + orow * a.ldo + c0;
                        if (full && (a.ldo & 3) == 0) *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (c0 + q < a.cout) op[q] = v[q];
                        }
                    } else {
                        T* op = reinterpret_cast<T*>(a.out) + orow * a.ldo + c0;
                        if (full && (a.ldo & 3) == 0 && sizeof(T) == 2) *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                        else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) if (c0 + q < a.cout) TT<T>::st(op + q, v[q]);
                        }
                    }
                }
            }
        }
    }
}

template <typename T, int S>
int conv_launch(const ConvArgs& a, hipStream_t st) {
    constexpr int SZ = TT<T>::SZ;
    constexpr int PY = (CT_Y - 1) * S + 3, PX = (CT_X - 1) * S + 3;
    const size_t lds = (size_t)PY * PX * (64 * SZ + 16);
    FW_SET_LDS_ONCE((conv3x3_kernel<T, S>), lds);
    const long ntile = (long)a.B * ((a.Ho + CT_Y - 1) / CT_Y) * ((a.Wo + CT_X - 1) / CT_X);
    const int grid = (int)(ntile < 2048 ? ntile : 2048);
    hipLaunchKernelGGL((conv3x3_kernel<T, S>), dim3(grid), dim3(256), lds, st, a);
    FW_LAUNCH_RET();
}

// =====================================================================================================
// explicit im2col / col2im of a 3x3 (or 1x1 = centre tap) convolution, stride S, padding 1 -- weight gradients, stride-2 input gradients
// =====================================================================================================
// col[(b, oy, ox)][tap * C + c] = x[(b, oy*S - 1 + ky, ox*S - 1 + kx)][c]  (0 outside)
template <typename T>
__global__ void im2col3_kernel(const T* __restrict__ x, long ldx, T* __restrict__ col, int B, int H, int W, int Ho, int Wo, int C, int S) {
    constexpr int E = TT<T>::E16;
    const int cv = C / E;
    const long n = (long)B * Ho * Wo * 9 * cv;
    for (long i = gtid(); i < n; i += gstride()) {
        const int g = (int)(i % cv);
        long t = i / cv;
        const int tap = (int)(t % 9); t /= 9;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int iy = oy * S - 1 + tap / 3, ix = ox * S - 1 + tap % 3;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        const uint4 v = *reinterpret_cast<const uint4*>(x + (((long)b * H + cy) * W + cx) * ldx + g * E);
        *reinterpret_cast<uint4*>(col + (i / cv) * (long)C + (long)g * E - (long)0) = ok ? v : make_uint4(0, 0, 0, 0);
    }
}
// dx[(b, y, x)][c] = sum over taps whose source pixel is an output pixel:  dcol[(b, oy, ox)][tap * C + c],  oy*S - 1 + ky = y
template <typename T>
__global__ void col2im3_kernel(const T* __restrict__ dcol, T* __restrict__ dx, long lddx, int B, int H, int W, int Ho, int Wo, int C, int S) {
    constexpr int E = TT<T>::E16;
    const int cv = C / E;
    const long n = (long)B * H * W * cv;
    for (long i = gtid(); i < n; i += gstride()) {
        const int g = (int)(i % cv);
        long t = i / cv;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int b = (int)(t / H);
        float s[E];
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int ny = yy + 1 - tap / 3, nx = xx + 1 - tap % 3;
            if (ny < 0 || nx < 0 || ny % S || nx % S) continue;
            const int oy = ny / S, ox = nx / S;
            if (oy >= Ho || ox >= Wo) continue;
            float f[E];
            load_vec<T>(dcol + (((long)b * Ho + oy) * Wo + ox) * 9 * C + (long)tap * C + g * E, f);
#pragma unroll
            for (int e = 0; e < E; ++e) s[e] += f[e];
        }
        store_vec<T>(dx + (((long)b * H + yy) * W + xx) * lddx + g * E, s);
    }
}

// =====================================================================================================
// DCNv2: modulated, bilinear im2col and its backward
// =====================================================================================================
// om: f32 [M][32] raw output of conv_offset_mask: channels 0..17 = offsets (2k = dy, 2k+1 = dx of tap k: deform_conv.py:59-61 re-joins
// o1 | o2 in order), 18..26 = mask logits.   col[p][k * C + c] = sigmoid(om[18 + k]) * bilinear(x[:, c], p + p_k + (dy, dx))
struct Bil { int y0, x0; float wy, wx; bool v00, v01, v10, v11; };
FW_DEV Bil bil_setup(float py, float px, int H, int W) {
    Bil q;
    const float fy = floorf(py), fx = floorf(px);
    q.y0 = (int)fy; q.x0 = (int)fx; q.wy = py - fy; q.wx = px - fx;
    const bool y0 = q.y0 >= 0 && q.y0 < H, y1 = q.y0 + 1 >= 0 && q.y0 + 1 < H;
    const bool x0 = q.x0 >= 0 && q.x0 < W, x1 = q.x0 + 1 >= 0 && q.x0 + 1 < W;
    q.v00 = y0 && x0; q.v01 = y0 && x1; q.v10 = y1 && x0; q.v11 = y1 && x1;
    return q;
}
template <typename T>
__global__ void dcn_im2col_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ om, T* __restrict__ col, int B, int H, int W, int C) {
    constexpr int E = TT<T>::E16;
    const int cv = C / E;
    const long n = (long)B * H * W * 9 * cv;
    for (long i = gtid(); i < n; i += gstride()) {
        const int g = (int)(i % cv);
        long t = i / cv;
        const int k = (int)(t % 9);
        const long p = t / 9;
        const int xx = (int)(p % W), yy = (int)((p / W) % H);
        const long b0 = (p / ((long)W * H)) * H * W;
        const float* o = om + p * 32;
        const float m = 1.0f / (1.0f + __expf(-o[18 + k]));
        const Bil q = bil_setup(yy + k / 3 - 1 + o[2 * k], xx + k % 3 - 1 + o[2 * k + 1], H, W);
        float s[E], f[E];
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = 0.f;
        const int cy0 = min(max(q.y0, 0), H - 1), cy1 = min(max(q.y0 + 1, 0), H - 1);
        const int cx0 = min(max(q.x0, 0), W - 1), cx1 = min(max(q.x0 + 1, 0), W - 1);
        const float w00 = q.v00 ? (1 - q.wy) * (1 - q.wx) : 0.f, w01 = q.v01 ? (1 - q.wy) * q.wx : 0.f;
        const float w10 = q.v10 ? q.wy * (1 - q.wx) : 0.f, w11 = q.v11 ? q.wy * q.wx : 0.f;
        load_vec<T>(x + (b0 + (long)cy0 * W + cx0) * ldx + g * E, f);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += w00 * f[e];
        load_vec<T>(x + (b0 + (long)cy0 * W + cx1) * ldx + g * E, f);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += w01 * f[e];
        load_vec<T>(x + (b0 + (long)cy1 * W + cx0) * ldx + g * E, f);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += w10 * f[e];
        load_vec<T>(x + (b0 + (long)cy1 * W + cx1) * ldx + g * E, f);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = (s[e] + w11 * f[e]) * m;
        store_vec<T>(col + t * C + (long)g * E, s);
    }
}
// One wave per (pixel, tap): lanes walk the channels.  dx (f32, pre-zeroed) += bilinear scatter; dom[p][2k], [2k+1], [18+k] = gradients
// of the raw offsets / mask logit (every (p, k) has exactly one writer).
template <typename T>
__global__ void dcn_bwd_kernel(const T* __restrict__ dcol, const T* __restrict__ x, long ldx, const float* __restrict__ om,
                               float* __restrict__ dx, long lddx, float* __restrict__ dom, int B, int H, int W, int C) {
    const long nw = (long)B * H * W * 9;
    const int l = lane_id();
    for (long t = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < nw; t += (long)gridDim.x * (blockDim.x >> 6)) {
        const int k = (int)(t % 9);
        const long p = t / 9;
        const int xx = (int)(p % W), yy = (int)((p / W) % H);
        const long b0 = (p / ((long)W * H)) * H * W;
        const float* o = om + p * 32;
        const float m = 1.0f / (1.0f + __expf(-o[18 + k]));
        const Bil q = bil_setup(yy + k / 3 - 1 + o[2 * k], xx + k % 3 - 1 + o[2 * k + 1], H, W);
        const int cy0 = min(max(q.y0, 0), H - 1), cy1 = min(max(q.y0 + 1, 0), H - 1);
        const int cx0 = min(max(q.x0, 0), W - 1), cx1 = min(max(q.x0 + 1, 0), W - 1);
        const long r00 = b0 + (long)cy0 * W + cx0, r01 = b0 + (long)cy0 * W + cx1, r10 = b0 + (long)cy1 * W + cx0, r11 = b0 + (long)cy1 * W + cx1;
        float sm = 0.f, sy = 0.f, sx = 0.f;
        for (int c = l; c < C; c += 64) {
            const float g = TT<T>::ld(dcol + t * C + c);
            const float a00 = q.v00 ? TT<T>::ld(x + r00 * ldx + c) : 0.f, a01 = q.v01 ? TT<T>::ld(x + r01 * ldx + c) : 0.f;
            const float a10 = q.v10 ? TT<T>::ld(x + r10 * ldx + c) : 0.f, a11 = q.v11 ? TT<T>::ld(x + r11 * ldx + c) : 0.f;
            const float val = (1 - q.wy) * ((1 - q.wx) * a00 + q.wx * a01) + q.wy * ((1 - q.wx) * a10 + q.wx * a11);
            sm += g * val;
            sy += g * m * ((1 - q.wx) * (a10 - a00) + q.wx * (a11 - a01));
            sx += g * m * ((1 - q.wy) * (a01 - a00) + q.wy * (a11 - a10));
            const float gm = g * m;
            if (q.v00) atomicAdd(dx + r00 * lddx + c, gm * (1 - q.wy) * (1 - q.wx));
            if (q.v01) atomicAdd(dx + r01 * lddx + c, gm * (1 - q.wy) * q.wx);
            if (q.v10) atomicAdd(dx + r10 * lddx + c, gm * q.wy * (1 - q.wx));
            if (q.v11) atomicAdd(dx + r11 * lddx + c, gm * q.wy * q.wx);
        }
        sm = wave_sum(sm); sy = wave_sum(sy); sx = wave_sum(sx);
        if (l == 0) {
            dom[p * 32 + 2 * k] = sy;
            dom[p * 32 + 2 * k + 1] = sx;
            dom[p * 32 + 18 + k] = sm * m * (1.0f - m);
        }
    }
}

// =====================================================================================================
// BatchNorm2d on token-major maps, elementwise pieces
// =====================================================================================================
// sums[0][c] += sum_rows x, sums[1][c] += sum_rows x^2   (block partials in registers, one atomic per column and block)
template <typename T>
__global__ void colstats_kernel(const T* __restrict__ x, long ldx, long rows, int C, float* __restrict__ sums) {
    // thread -> (column group of 4, row lane); a block covers all column groups x RL row lanes
    const int cg = C / 4;
    const int col = threadIdx.x % cg, rl = threadIdx.x / cg, RL = blockDim.x / cg;
    if (rl >= RL) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    for (long r = (long)blockIdx.x * RL + rl; r < rows; r += (long)gridDim.x * RL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float v = TT<T>::ld(x + r * ldx + col * 4 + e); s[e] += v; q[e] += v * v; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { atomicAdd(sums + col * 4 + e, s[e]); atomicAdd(sums + C + col * 4 + e, q[e]); }
}
// mr[0][c] = mean, mr[1][c] = rstd; train: from the sums, and fold them into the running statistics (momentum, unbiased variance)
__global__ void bn_finalize_kernel(const float* sums, long rows, float* rmean, float* rvar, long long* nbt, float* mr, int C, int training,
                                   float eps, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && training && nbt) *nbt += 1;
    if (c >= C) return;
    if (training) {
        const float mean = sums[c] / rows;
        const float var = fmaxf(sums[C + c] / rows - mean * mean, 0.f);
        mr[c] = mean; mr[C + c] = rsqrtf(var + eps);
        rmean[c] = rmean[c] * (1.f - momentum) + momentum * mean;
        rvar[c] = rvar[c] * (1.f - momentum) + momentum * var * ((float)rows / (float)(rows > 1 ? rows - 1 : 1));
    } else {
        mr[c] = rmean[c]; mr[C + c] = rsqrtf(rvar[c] + eps);
    }
}
// y = act( (x - mean) * rstd * gamma + beta [+ res] )
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ mr, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ res, long ldr, T* __restrict__ y, long ldy, long rows,
                                int C, float slope) {
    const int cg = C / 4;
    const long n = rows * cg;
    for (long i = gtid(); i < n; i += gstride()) {
        const int c0 = (int)(i % cg) * 4;
        const long r = i / cg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            float v = (TT<T>::ld(x + r * ldx + c) - mr[c]) * mr[C + c] * gamma[c] + beta[c];
            if (res) v += TT<T>::ld(res + r * ldr + c);
            TT<T>::st(y + r * ldy + c, lrelu_f(v, slope));
        }
    }
}
// backward pass 1: dz = dy * lrelu'(y);  sums[0][c] += sum dz, sums[1][c] += sum dz * xhat
template <typename T>
__global__ void bn_bwd_stats_kernel(const T* __restrict__ dy, long ldd, const T* __restrict__ y, long ldy, const T* __restrict__ x, long ldx,
                                    const float* __restrict__ mr, float slope, long rows, int C, float* __restrict__ sums) {
    const int cg = C / 4;
    const int col = threadIdx.x % cg, rl = threadIdx.x / cg, RL = blockDim.x / cg;
    if (rl >= RL) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    for (long r = (long)blockIdx.x * RL + rl; r < rows; r += (long)gridDim.x * RL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = col * 4 + e;
            float dz = TT<T>::ld(dy + r * ldd + c);
            if (slope != 1.0f && TT<T>::ld(y + r * ldy + c) < 0.f) dz *= slope;
            s[e] += dz; q[e] += dz * (TT<T>::ld(x + r * ldx + c) - mr[c]) * mr[C + c];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { atomicAdd(sums + col * 4 + e, s[e]); atomicAdd(sums + C + col * 4 + e, q[e]); }
}
// backward pass 2: dx = gamma * rstd * (dz - s0 / M - xhat * s1 / M) (train) | gamma * rstd * dz (eval);  dres = dz (optional)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, long ldd, const T* __restrict__ y, long ldy, const T* __restrict__ x, long ldx,
                                    const float* __restrict__ mr, const float* __restrict__ gamma, const float* __restrict__ sums,
                                    float slope, int training, T* __restrict__ dx, long lddx, T* __restrict__ dres, long lddr, long rows, int C) {
    const int cg = C / 4;
    const long n = rows * cg;
    const float inv = 1.0f / (float)rows;
    for (long i = gtid(); i < n; i += gstride()) {
        const int c0 = (int)(i % cg) * 4;
        const long r = i / cg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + e;
            float dz = TT<T>::ld(dy + r * ldd + c);
            if (slope != 1.0f && TT<T>::ld(y + r * ldy + c) < 0.f) dz *= slope;
            if (dres) TT<T>::st(dres + r * lddr + c, dz);
            const float xh = (TT<T>::ld(x + r * ldx + c) - mr[c]) * mr[C + c];
            const float g = gamma[c] * mr[C + c];
            TT<T>::st(dx + r * lddx + c, training ? g * (dz - sums[c] * inv - xh * sums[C + c] * inv) : g * dz);
        }
    }
}
// y = lrelu(x, slope) on T;  backward dx = dy * (y < 0 ? slope : 1)
template <typename T>
__global__ void lrelu_t_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long rows, int C, float slope) {
    const long n = rows * C;
    for (long i = gtid(); i < n; i += gstride()) TT<T>::st(y + (i / C) * ldy + i % C, lrelu_f(TT<T>::ld(x + (i / C) * ldx + i % C), slope));
}
template <typename T>
__global__ void lrelu_t_bwd_kernel(const T* __restrict__ dy, long ldd, const T* __restrict__ y, long ldy, T* __restrict__ dx, long lddx, long rows,
                                   int C, float slope) {
    const long n = rows * C;
    for (long i = gtid(); i < n; i += gstride()) {
        const long r = i / C; const int c = (int)(i % C);
        const float g = TT<T>::ld(dy + r * ldd + c);
        TT<T>::st(dx + r * lddx + c, TT<T>::ld(y + r * ldy + c) < 0.f ? g * slope : g);
    }
}
// DGM + the LeakyReLU that follows it in DGB (decoder_DGRN.py:22-32,79-81): out = lrelu(x + dcn + x * gamma + beta)
template <typename T>
__global__ void dgm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ dcn, const T* __restrict__ gamma, const T* __restrict__ beta,
                               T* __restrict__ out, long n, float slope) {
    for (long i = gtid(); i < n; i += gstride()) {
        const float xv = TT<T>::ld(x + i);
        TT<T>::st(out + i, lrelu_f(xv + TT<T>::ld(dcn + i) + xv * TT<T>::ld(gamma + i) + TT<T>::ld(beta + i), slope));
    }
}
// dz = dout * lrelu'(out);  dx = dz (1 + gamma);  dgamma = dz x;  dz itself is both d(dcn) and d(beta)
template <typename T>
__global__ void dgm_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ gamma,
                               T* __restrict__ dx, T* __restrict__ dz_out, T* __restrict__ dgamma, long n, float slope) {
    for (long i = gtid(); i < n; i += gstride()) {
        float dz = TT<T>::ld(dout + i);
        if (TT<T>::ld(out + i) < 0.f) dz *= slope;
        TT<T>::st(dz_out + i, dz);
        TT<T>::st(dx + i, dz * (1.0f + TT<T>::ld(gamma + i)));
        TT<T>::st(dgamma + i, dz * TT<T>::ld(x + i));
    }
}
// global average pool over the P pixels of an image: [B * P][C] T -> f32 [B][C]; backward broadcasts dgap / P
template <typename T>
__global__ void gap_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, int P, int C) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += TT<T>::ld(x + ((long)b * P + p) * ldx + c);
        out[(long)b * C + c] = s / P;
    }
}
template <typename T>
__global__ void gap_bwd_kernel(const float* __restrict__ dgap, T* __restrict__ dx, long lddx, long rows, int P, int C) {
    const long n = rows * C;
    for (long i = gtid(); i < n; i += gstride()) {
        const long r = i / C; const int c = (int)(i % C);
        TT<T>::st(dx + r * lddx + c, dgap[(r / P) * C + c] / P);
    }
}
// image planes f32 [B][Ci][H][W] <-> token-major T [B*H*W][ld] (channels >= Ci zero-filled up to Cp)
template <typename T>
__global__ void nchw_to_tokens_kernel(const float* __restrict__ img, T* __restrict__ tok, long ld, int B, int Ci, int HW, int Cp) {
    const long n = (long)B * HW * Cp;
    for (long i = gtid(); i < n; i += gstride()) {
        const int c = (int)(i % Cp);
        const long r = i / Cp;
        TT<T>::st(tok + r * ld + c, c < Ci ? img[((r / HW) * Ci + c) * HW + r % HW] : 0.f);
    }
}
template <typename T>
__global__ void tokens_to_nchw_kernel(const T* __restrict__ tok, long ld, float* __restrict__ img, int B, int Ci, int HW) {
    const long n = (long)B * Ci * HW;
    for (long i = gtid(); i < n; i += gstride()) {
        const long p = i % HW; const int c = (int)((i / HW) % Ci); const long b = i / ((long)HW * Ci);
        img[i] = TT<T>::ld(tok + (b * HW + p) * ld + c);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)
#define LAUNCH(kern, n, ...)                                                           \
    do {                                                                               \
        hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(TPB), 0, ST, __VA_ARGS__);    \
        FW_LAUNCH_RET();                                                               \
    } while (0)

// 3x3 (taps = 0x1ff) or 1x1 (taps = 0x010, weights still laid out [coutp][9 * cin]) convolution, padding 1, stride 1 | 2, as an
// implicit GEMM.  x / x2: T [B*H*W][ld], channels [0, cin1) from x and [cin1, cin) from x2 (x2 may be null, then cin1 = cin).
// w: T [roundup(cout, 16)][9 * cin], element (co, tap, ci).  out: T or f32 [B*Ho*Wo][ldo].  act: 0 | 1 (LeakyReLU slope).  res: T.
extern "C" int fw_conv3x3(int dtype, const void* x, long ldx, const void* x2, long ldx2, int cin, int cin1, const void* w,
                          const float* bias, void* out, long ldo, int out_f32, const void* res, long ldr, int cout, int B, int H,
                          int W, int stride, int taps, int act, float slope, void* stream) {
    const int sz = dtype == FW_DT_BF16 ? 2 : 4, ce = 64 / sz;
    FW_CHECK_ARG(dtype == FW_DT_F32 || dtype == FW_DT_BF16);
    FW_CHECK_ARG(x && w && out && B > 0 && H > 0 && W > 0 && cout > 0 && (stride == 1 || stride == 2));
    FW_CHECK_ARG(cin > 0 && cin % ce == 0 && cin1 > 0 && cin1 <= cin && (cin1 == cin || (x2 && cin1 % 64 == 0)));
    FW_CHECK_ARG((ldx * sz) % 16 == 0 && ldx >= cin1 && (!x2 || ((ldx2 * sz) % 16 == 0 && ldx2 >= cin - cin1)));
    FW_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && (!x2 || ((uintptr_t)x2 & 15) == 0));
    FW_CHECK_ARG((taps & 0x1ff) != 0 && ldo >= cout && (!res || ldr >= cout));
    ConvArgs a = {};
    a.x = (const char*)x; a.ldx = ldx; a.x2 = (const char*)x2; a.ldx2 = ldx2; a.cin = cin; a.cin1 = x2 ? cin1 : cin;
    a.w = (const char*)w; a.bias = bias; a.out = (char*)out; a.ldo = ldo; a.out_f32 = (out_f32 || dtype == FW_DT_F32) ? 1 : 0;
    a.res = (const char*)res; a.ldr = ldr; a.cout = cout; a.B = B; a.H = H; a.W = W; a.stride = stride;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1; a.taps = taps & 0x1ff; a.act = act; a.slope = slope;
    if (dtype == FW_DT_BF16) return stride == 1 ? conv_launch<bf16raw, 1>(a, ST) : conv_launch<bf16raw, 2>(a, ST);
    return stride == 1 ? conv_launch<float, 1>(a, ST) : conv_launch<float, 2>(a, ST);
}
extern "C" int fw_im2col3(int dtype, const void* x, long ldx, void* col, int B, int H, int W, int C, int stride, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(x && col && C % e == 0 && ldx % e == 0 && (stride == 1 || stride == 2));
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long n = (long)B * Ho * Wo * 9 * (C / e);
    if (dtype == FW_DT_BF16) LAUNCH((im2col3_kernel<bf16raw>), n, (const bf16raw*)x, ldx, (bf16raw*)col, B, H, W, Ho, Wo, C, stride);
    LAUNCH((im2col3_kernel<float>), n, (const float*)x, ldx, (float*)col, B, H, W, Ho, Wo, C, stride);
}
extern "C" int fw_col2im3(int dtype, const void* dcol, void* dx, long lddx, int B, int H, int W, int C, int stride, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(dcol && dx && C % e == 0 && lddx % e == 0 && (stride == 1 || stride == 2));
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long n = (long)B * H * W * (C / e);
    if (dtype == FW_DT_BF16) LAUNCH((col2im3_kernel<bf16raw>), n, (const bf16raw*)dcol, (bf16raw*)dx, lddx, B, H, W, Ho, Wo, C, stride);
    LAUNCH((col2im3_kernel<float>), n, (const float*)dcol, (float*)dx, lddx, B, H, W, Ho, Wo, C, stride);
}
extern "C" int fw_dcn_im2col(int dtype, const void* x, long ldx, const float* om, void* col, int B, int H, int W, int C, void* stream) {
    const int e = dtype == FW_DT_BF16 ? 8 : 4;
    FW_CHECK_ARG(x && om && col && C % e == 0 && ldx % e == 0);
    const long n = (long)B * H * W * 9 * (C / e);
    if (dtype == FW_DT_BF16) LAUNCH((dcn_im2col_kernel<bf16raw>), n, (const bf16raw*)x, ldx, om, (bf16raw*)col, B, H, W, C);
    LAUNCH((dcn_im2col_kernel<float>), n, (const float*)x, ldx, om, (float*)col, B, H, W, C);
}
// dx: f32 [M][lddx], ACCUMULATED into (pre-zero it);  dom: f32 [M][32], every offset / mask entry written.
extern "C" int fw_dcn_bwd(int dtype, const void* dcol, const void* x, long ldx, const float* om, float* dx, long lddx, float* dom,
                          int B, int H, int W, int C, void* stream) {
    FW_CHECK_ARG(dcol && x && om && dx && dom && C > 0);
    const long nw = (long)B * H * W * 9;
    const int grid = (int)((nw + 3) / 4 < 65536 ? (nw + 3) / 4 : 65536);
    if (dtype == FW_DT_BF16)
        hipLaunchKernelGGL((dcn_bwd_kernel<bf16raw>), dim3(grid), dim3(256), 0, ST, (const bf16raw*)dcol, (const bf16raw*)x, ldx, om, dx, lddx, dom, B, H, W, C);
    else
        hipLaunchKernelGGL((dcn_bwd_kernel<float>), dim3(grid), dim3(256), 0, ST, (const float*)dcol, (const float*)x, ldx, om, dx, lddx, dom, B, H, W, C);
    FW_LAUNCH_RET();
}
// BatchNorm2d over the rows of a token-major map.  sums: f32 [2][C] scratch, ZEROED by the caller.  mr: f32 [2][C] (mean, rstd) out.
extern "C" int fw_bn_cl_fwd(int dtype, const void* x, long ldx, const float* gamma, const float* beta, float* rmean, float* rvar,
                            long long* nbt, float* sums, float* mr, const void* res, long ldr, void* y, long ldy, long rows, int C,
                            int training, float eps, float momentum, float slope, void* stream) {
    FW_CHECK_ARG(x && gamma && beta && rmean && rvar && sums && mr && y && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024);
    const int cg = C / 4, RL = 256 / cg > 0 ? 256 / cg : 1;
    if (training) {
        const int grid = (int)((rows + RL - 1) / RL < 1024 ? (rows + RL - 1) / RL : 1024);
        if (dtype == FW_DT_BF16) hipLaunchKernelGGL((colstats_kernel<bf16raw>), dim3(grid), dim3(256), 0, ST, (const bf16raw*)x, ldx, rows, C, sums);
        else hipLaunchKernelGGL((colstats_kernel<float>), dim3(grid), dim3(256), 0, ST, (const float*)x, ldx, rows, C, sums);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, sums, rows, rmean, rvar, nbt, mr, C, training, eps, momentum);
    const long n = rows * cg;
    if (dtype == FW_DT_BF16)
        LAUNCH((bn_apply_kernel<bf16raw>), n, (const bf16raw*)x, ldx, mr, gamma, beta, (const bf16raw*)res, ldr, (bf16raw*)y, ldy, rows, C, slope);
    LAUNCH((bn_apply_kernel<float>), n, (const float*)x, ldx, mr, gamma, beta, (const float*)res, ldr, (float*)y, ldy, rows, C, slope);
}
// sums: f32 [2][C], ZEROED by the caller; on return sums[0] = d(beta), sums[1] = d(gamma).  dres (optional) receives dy * lrelu'(y).
extern "C" int fw_bn_cl_bwd(int dtype, const void* dy, long ldd, const void* y, long ldy, const void* x, long ldx, const float* mr,
                            const float* gamma, float* sums, void* dx, long lddx, void* dres, long lddr, long rows, int C, int training,
                            float slope, void* stream) {
    FW_CHECK_ARG(dy && y && x && mr && gamma && sums && dx && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024);
    const int cg = C / 4, RL = 256 / cg > 0 ? 256 / cg : 1;
    const int grid = (int)((rows + RL - 1) / RL < 1024 ? (rows + RL - 1) / RL : 1024);
    const long n = rows * cg;
    if (dtype == FW_DT_BF16) {
        hipLaunchKernelGGL((bn_bwd_stats_kernel<bf16raw>), dim3(grid), dim3(256), 0, ST, (const bf16raw*)dy, ldd, (const bf16raw*)y, ldy,
                           (const bf16raw*)x, ldx, mr, slope, rows, C, sums);
        LAUNCH((bn_bwd_apply_kernel<bf16raw>), n, (const bf16raw*)dy, ldd, (const bf16raw*)y, ldy, (const bf16raw*)x, ldx, mr, gamma, sums, slope,
               training, (bf16raw*)dx, lddx, (bf16raw*)dres, lddr, rows, C);
    }
    hipLaunchKernelGGL((bn_bwd_stats_kernel<float>), dim3(grid), dim3(256), 0, ST, (const float*)dy, ldd, (const float*)y, ldy, (const float*)x, ldx,
                       mr, slope, rows, C, sums);
    LAUNCH((bn_bwd_apply_kernel<float>), n, (const float*)dy, ldd, (const float*)y, ldy, (const float*)x, ldx, mr, gamma, sums, slope, training,
           (float*)dx, lddx, (float*)dres, lddr, rows, C);
}
extern "C" int fw_lrelu_t(int dtype, const void* x, long ldx, void* y, long ldy, long rows, int C, float slope, void* stream) {
    FW_CHECK_ARG(x && y && rows > 0 && C > 0);
    if (dtype == FW_DT_BF16) LAUNCH((lrelu_t_kernel<bf16raw>), rows * C, (const bf16raw*)x, ldx, (bf16raw*)y, ldy, rows, C, slope);
    LAUNCH((lrelu_t_kernel<float>), rows * C, (const float*)x, ldx, (float*)y, ldy, rows, C, slope);
}
extern "C" int fw_lrelu_t_bwd(int dtype, const void* dy, long ldd, const void* y, long ldy, void* dx, long lddx, long rows, int C, float slope,
                              void* stream) {
    FW_CHECK_ARG(dy && y && dx && rows > 0 && C > 0);
    if (dtype == FW_DT_BF16) LAUNCH((lrelu_t_bwd_kernel<bf16raw>), rows * C, (const bf16raw*)dy, ldd, (const bf16raw*)y, ldy, (bf16raw*)dx, lddx, rows, C, slope);
    LAUNCH((lrelu_t_bwd_kernel<float>), rows * C, (const float*)dy, ldd, (const float*)y, ldy, (float*)dx, lddx, rows, C, slope);
}
// contiguous [n] T tensors
extern "C" int fw_dgm_fwd(int dtype, const void* x, const void* dcn, const void* gamma, const void* beta, void* out, long n, float slope,
                          void* stream) {
    FW_CHECK_ARG(x && dcn && gamma && beta && out && n > 0);
    if (dtype == FW_DT_BF16) LAUNCH((dgm_fwd_kernel<bf16raw>), n, (const bf16raw*)x, (const bf16raw*)dcn, (const bf16raw*)gamma, (const bf16raw*)beta, (bf16raw*)out, n, slope);
    LAUNCH((dgm_fwd_kernel<float>), n, (const float*)x, (const float*)dcn, (const float*)gamma, (const float*)beta, (float*)out, n, slope);
}
extern "C" int fw_dgm_bwd(int dtype, const void* dout, const void* out, const void* x, const void* gamma, void* dx, void* dz, void* dgamma,
                          long n, float slope, void* stream) {
    FW_CHECK_ARG(dout && out && x && gamma && dx && dz && dgamma && n > 0);
    if (dtype == FW_DT_BF16)
        LAUNCH((dgm_bwd_kernel<bf16raw>), n, (const bf16raw*)dout, (const bf16raw*)out, (const bf16raw*)x, (const bf16raw*)gamma, (bf16raw*)dx, (bf16raw*)dz, (bf16raw*)dgamma, n, slope);
    LAUNCH((dgm_bwd_kernel<float>), n, (const float*)dout, (const float*)out, (const float*)x, (const float*)gamma, (float*)dx, (float*)dz, (float*)dgamma, n, slope);
}
extern "C" int fw_gap_cl(int dtype, const void* x, long ldx, float* out, int B, int P, int C, void* stream) {
    FW_CHECK_ARG(x && out && B > 0 && P > 0 && C > 0);
    if (dtype == FW_DT_BF16) hipLaunchKernelGGL((gap_kernel<bf16raw>), dim3(B), dim3(256), 0, ST, (const bf16raw*)x, ldx, out, P, C);
    else hipLaunchKernelGGL((gap_kernel<float>), dim3(B), dim3(256), 0, ST, (const float*)x, ldx, out, P, C);
    FW_LAUNCH_RET();
}
extern "C" int fw_gap_cl_bwd(int dtype, const float* dgap, void* dx, long lddx, int B, int P, int C, void* stream) {
    FW_CHECK_ARG(dgap && dx && B > 0 && P > 0 && C > 0);
    const long rows = (long)B * P;
    if (dtype == FW_DT_BF16) LAUNCH((gap_bwd_kernel<bf16raw>), rows * C, dgap, (bf16raw*)dx, lddx, rows, P, C);
    LAUNCH((gap_bwd_kernel<float>), rows * C, dgap, (float*)dx, lddx, rows, P, C);
}
extern "C" int fw_nchw_to_tokens(int dtype, const float* img, void* tok, long ld, int B, int Ci, int HW, int Cp, void* stream) {
    FW_CHECK_ARG(img && tok && B > 0 && Ci > 0 && HW > 0 && Cp >= Ci && ld >= Cp);
    const long n = (long)B * HW * Cp;
    if (dtype == FW_DT_BF16) LAUNCH((nchw_to_tokens_kernel<bf16raw>), n, img, (bf16raw*)tok, ld, B, Ci, HW, Cp);
    LAUNCH((nchw_to_tokens_kernel<float>), n, img, (float*)tok, ld, B, Ci, HW, Cp);
}
extern "C" int fw_tokens_to_nchw(int dtype, const void* tok, long ld, float* img, int B, int Ci, int HW, void* stream) {
    FW_CHECK_ARG(img && tok && B > 0 && Ci > 0 && HW > 0 && ld >= Ci);
    const long n = (long)B * Ci * HW;
    if (dtype == FW_DT_BF16) LAUNCH((tokens_to_nchw_kernel<bf16raw>), n, (const bf16raw*)tok, ld, img, B, Ci, HW);
    LAUNCH((tokens_to_nchw_kernel<float>), n, (const float*)tok, ld, img, B, Ci, HW);
}
