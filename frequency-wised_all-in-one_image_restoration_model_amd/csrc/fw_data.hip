// Callers either side of the model as device kernels, so that neither the input pipeline nor the evaluation
// bounds the training rate:
//   * fw_train_batch  -- one launch makes a whole training batch from uint8 images resident in HBM: Gaussian noise synthesis on the
//     uint8 grid (utils/dataset_utils.py:122-126), two independent random crops of the SAME degraded image (:131-132, `_crop_patch`),
//     one of the 7 flip / rotation modes per crop (utils/image_utils.py:133-182 `random_augmentation`), ToTensor scaling (:134-135).
//     No host round trip: crop origins and modes come from one device tensor of random integers, the noise is counter-based
//     (fw_common.h hash -> Box-Muller), so both crops see the same noisy image although that image is never materialised.
//   * fw_tile_gather / fw_tile_blend -- test.py:47-71: cut a test image into tiles (stride = tile, last tile flush with the border),
//     and average the RESTORED tiles over their overlap (gather form: at most two tiles per axis cover a pixel; no atomics).
//   * fw_ssim7 -- structural similarity as utils/val_utils.py:50-66 calls it (skimage.metrics.structural_similarity defaults: 7x7
//     uniform window, K1 = 0.01, K2 = 0.03, sample covariance, border of 3 pixels cropped, mean over channels), inputs clipped to [0, 1].
// All HBM-bound byte / float streaming work: grid-stride kernels, coalesced along x.
#include "fw_common.h"

namespace {
constexpr int TPB = 256;
FW_DEV long gtid() { return (long)blockIdx.x * blockDim.x + threadIdx.x; }
FW_DEV long gstride() { return (long)gridDim.x * blockDim.x; }
static inline int grid_for(long n) { long g = (n + TPB - 1) / TPB; return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g)); }

// ---- training batch ---------------------------------------------------------------------------------------------------------------
// tab[b] = {clean ptr (u8 [3][H][W]), degraded ptr (u8, same shape) or 0, H, W};  rnd[b] = {ry1, rx1, rm1, ry2, rx2, rm2} >= 0
// out: d1, d2, c1, c2 f32 [B][3][S][S].   sigma[b]: noise level when no degraded image is given (0: the clean image itself).
__global__ void train_batch_kernel(const long long* __restrict__ tab, const int* __restrict__ rnd, const float* __restrict__ sigma,
                                   const unsigned* __restrict__ seed, unsigned site, float* __restrict__ d1, float* __restrict__ d2,
                                   float* __restrict__ c1, float* __restrict__ c2, int B, int S) {
    const long per = (long)3 * S * S, n = (long)B * 2 * per;
    for (long i = gtid(); i < n; i += gstride()) {
        const int ox = (int)(i % S), oy = (int)((i / S) % S), c = (int)((i / ((long)S * S)) % 3);
        const int v = (int)((i / per) % 2), b = (int)(i / (2 * per));
        const unsigned char* clean = reinterpret_cast<const unsigned char*>(tab[b * 4 + 0]);
        const unsigned char* degr = reinterpret_cast<const unsigned char*>(tab[b * 4 + 1]);
        const int H = (int)tab[b * 4 + 2], W = (int)tab[b * 4 + 3];
        const int* r = rnd + b * 6 + v * 3;
        const int y0 = r[0] % (H - S + 1), x0 = r[1] % (W - S + 1), mode = 1 + r[2] % 7;
        const int k = mode >> 1, ii = (mode & 1) ? S - 1 - oy : oy, jj = ox;          // flipud last -> undone first
        int py, px;
        if (k == 0) { py = ii; px = jj; }
        else if (k == 1) { py = jj; px = S - 1 - ii; }
        else if (k == 2) { py = S - 1 - ii; px = S - 1 - jj; }
        else { py = S - 1 - jj; px = ii; }
        const long pi = ((long)c * H + (y0 + py)) * W + (x0 + px);
        const float g = (float)clean[pi];
        float d;
        if (degr) d = (float)degr[pi];
        else {
            const float sg = sigma[b];
            d = g;
            if (sg > 0.f) {
                const unsigned key = fw_site_key(seed[0], site + (unsigned)b);
                const unsigned r1 = fw_hash32((unsigned)pi ^ key), r2 = fw_hash32(r1 + 0x9E3779B9U);
                const float u1 = ((float)r1 + 1.0f) * 2.3283064365386963e-10f, u2 = (float)r2 * 2.3283064365386963e-10f;
                const float z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
                d = floorf(fminf(fmaxf(g + z * sg, 0.f), 255.f));                     // clip, then astype(uint8) truncates
            }
        }
        const long o = ((long)b * 3 + c) * S * S + (long)oy * S + ox;
        (v ? d2 : d1)[o] = __fdiv_rn(d, 255.0f);                                 // ToTensor divides (correctly rounded), bit-exact with numpy
        (v ? c2 : c1)[o] = __fdiv_rn(g, 255.0f);
    }
}

// ---- evaluation tiles -------------------------------------------------------------------------------------------------------------
__global__ void tile_gather_kernel(const float* __restrict__ img, const int* __restrict__ ys, const int* __restrict__ xs, float* __restrict__ tiles,
                                   int C, int H, int W, int ny, int nx, int T) {
    const long n = (long)ny * nx * C * T * T;
    for (long i = gtid(); i < n; i += gstride()) {
        const int j = (int)(i % T), ii = (int)((i / T) % T), c = (int)((i / ((long)T * T)) % C), t = (int)(i / ((long)T * T * C));
        tiles[i] = img[((long)c * H + ys[t / nx] + ii) * W + xs[t % nx] + j];
    }
}
__global__ void tile_blend_kernel(const float* __restrict__ tiles, const int* __restrict__ ys, const int* __restrict__ xs, float* __restrict__ out,
                                  int C, int H, int W, int ny, int nx, int T) {
    const long n = (long)C * H * W;
    for (long i = gtid(); i < n; i += gstride()) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((long)W * H));
        float acc = 0.f, cnt = 0.f;
        for (int a = 0; a < ny; ++a) {
            const int dy = y - ys[a];
            if (dy < 0 || dy >= T) continue;
            for (int b = 0; b < nx; ++b) {
                const int dx = x - xs[b];
                if (dx < 0 || dx >= T) continue;
                acc += tiles[(((long)(a * nx + b) * C + c) * T + dy) * T + dx];
                cnt += 1.f;
            }
        }
        out[i] = acc / cnt;
    }
}

// ---- SSIM -------------------------------------------------------------------------------------------------------------------------
// a, b: f32 [n][C][H][W]; out[i] += sum over (c, interior y, x) of the SSIM map of image i (the caller divides by C (H-6) (W-6))
__global__ void ssim7_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n, int C, int H, int W) {
    __shared__ float red[TPB / 64];
    const int hh = H - 6, ww = W - 6;
    const long per = (long)C * hh * ww;
    const int img = blockIdx.y;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ww), y = (int)((i / ww) % hh), c = (int)(i / ((long)ww * hh));
        const float* pa = a + (((long)img * C + c) * H + y) * W + x;
        const float* pb = b + (((long)img * C + c) * H + y) * W + x;
        float sa = 0.f, sb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
            for (int dx = 0; dx < 7; ++dx) {
                const float u = fminf(fmaxf(pa[dy * W + dx], 0.f), 1.f), v = fminf(fmaxf(pb[dy * W + dx], 0.f), 1.f);
                sa += u; sb += v; saa += u * u; sbb += v * v; sab += u * v;
            }
        const float inv = 1.0f / 49.0f, cn = 49.0f / 48.0f;
        const float ua = sa * inv, ub = sb * inv;
        const float va = cn * (saa * inv - ua * ua), vb = cn * (sbb * inv - ub * ub), vab = cn * (sab * inv - ua * ub);
        const float C1 = 1e-4f, C2 = 9e-4f;
        acc += ((2.f * ua * ub + C1) * (2.f * vab + C2)) / ((ua * ua + ub * ub + C1) * (va + vb + C2));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < TPB / 64; ++i) s += red[i];
        atomicAdd(out + img, s);
    }
}
}  // namespace

#define ST ((hipStream_t)stream)
extern "C" int fw_train_batch(const void* tab, const int* rnd, const float* sigma, const void* seed, int site, float* d1, float* d2, float* c1,
                              float* c2, int B, int S, void* stream) {
    FW_CHECK_ARG(tab && rnd && sigma && seed && d1 && d2 && c1 && c2 && B > 0 && S > 0);
    hipLaunchKernelGGL(train_batch_kernel, dim3(grid_for((long)B * 6 * S * S)), dim3(TPB), 0, ST, (const long long*)tab, rnd, sigma,
                       (const unsigned*)seed, (unsigned)site, d1, d2, c1, c2, B, S);
    FW_LAUNCH_RET();
}
extern "C" int fw_tile_gather(const float* img, const int* ys, const int* xs, float* tiles, int C, int H, int W, int ny, int nx, int T,
                              void* stream) {
    FW_CHECK_ARG(img && ys && xs && tiles && C > 0 && T > 0 && H >= T && W >= T && ny > 0 && nx > 0);
    hipLaunchKernelGGL(tile_gather_kernel, dim3(grid_for((long)ny * nx * C * T * T)), dim3(TPB), 0, ST, img, ys, xs, tiles, C, H, W, ny, nx, T);
    FW_LAUNCH_RET();
}
extern "C" int fw_tile_blend(const float* tiles, const int* ys, const int* xs, float* out, int C, int H, int W, int ny, int nx, int T,
                             void* stream) {
    FW_CHECK_ARG(tiles && ys && xs && out && C > 0 && T > 0 && H >= T && W >= T && ny > 0 && nx > 0);
    hipLaunchKernelGGL(tile_blend_kernel, dim3(grid_for((long)C * H * W)), dim3(TPB), 0, ST, tiles, ys, xs, out, C, H, W, ny, nx, T);
    FW_LAUNCH_RET();
}
extern "C" int fw_ssim7(const float* a, const float* b, float* out, int n, int C, int H, int W, void* stream) {
    FW_CHECK_ARG(a && b && out && n > 0 && C > 0 && H >= 7 && W >= 7);
    const long per = (long)C * (H - 6) * (W - 6);
    long gx = (per + TPB - 1) / TPB;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ssim7_kernel, dim3((unsigned)gx, (unsigned)n), dim3(TPB), 0, ST, a, b, out, n, C, H, W);
    FW_LAUNCH_RET();
}
