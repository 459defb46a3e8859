// Does the all-VALU direct-sum loop lose time to its LDS reads?  The compiled loop of k_direct<4> waits for each of the
// last four ds_read_b128 of an 8-unrolled trip right where it issues them (s_waitcnt lgkmcnt(0) behind the read).
// Variants: the library's loop; j-bodies fetched a chunk ahead into registers (double-buffered chunks of 4 / 8); more
// i-bodies per thread.  1 M bodies, 256-thread blocks.   hipcc --offload-arch=gfx950 -O3 -o direct_valu_sched direct_valu_sched.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int IB>
__device__ __forceinline__ void pair(const float4 q, const float (&px)[IB], const float (&py)[IB], const float (&pz)[IB], float eps2,
                                     float (&sx)[IB], float (&sy)[IB], float (&sz)[IB]) {
#pragma unroll
    for (int k = 0; k < IB; k++) {
        const float dx = q.x - px[k], dy = q.y - py[k], dz = q.z - pz[k];
        const float r2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
        const float inv = __builtin_amdgcn_rsqf(r2);
        const float f = q.w * inv * inv * inv;
        sx[k] = fmaf(f, dx, sx[k]); sy[k] = fmaf(f, dy, sy[k]); sz[k] = fmaf(f, dz, sz[k]);
    }
}

// MODE 0: the library's loop (unroll 8).  MODE 1: chunks of CH j-bodies, the next chunk's LDS reads issued before the
// current chunk's arithmetic.
template <int IB, int MODE, int CH>
__global__ __launch_bounds__(256) void k_direct(const float4 *__restrict__ posm, int n, float eps2, double *__restrict__ out) {
    __shared__ float4 tile[256];
    const int i0 = blockIdx.x * (256 * IB) + threadIdx.x;
    float px[IB], py[IB], pz[IB];
    double ax[IB], ay[IB], az[IB];
#pragma unroll
    for (int k = 0; k < IB; k++) {
        const int i = i0 + k * 256;
        const float4 p = i < n ? posm[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        px[k] = p.x; py[k] = p.y; pz[k] = p.z; ax[k] = ay[k] = az[k] = 0.0;
    }
    for (int t = 0; t < (n + 255) / 256; t++) {
        const int j = t * 256 + threadIdx.x;
        tile[threadIdx.x] = j < n ? posm[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        float sx[IB], sy[IB], sz[IB];
#pragma unroll
        for (int k = 0; k < IB; k++) sx[k] = sy[k] = sz[k] = 0.f;
        if (MODE == 0) {
#pragma unroll 8
            for (int jj = 0; jj < 256; jj++) pair<IB>(tile[jj], px, py, pz, eps2, sx, sy, sz);
        } else {
            float4 cur[CH], nxt[CH];
#pragma unroll
            for (int c = 0; c < CH; c++) cur[c] = tile[c];
#pragma unroll 1
            for (int jj = 0; jj < 256; jj += CH) {
                const int nb = jj + CH < 256 ? jj + CH : 0;
#pragma unroll
                for (int c = 0; c < CH; c++) nxt[c] = tile[nb + c];
#pragma unroll
                for (int c = 0; c < CH; c++) pair<IB>(cur[c], px, py, pz, eps2, sx, sy, sz);
#pragma unroll
                for (int c = 0; c < CH; c++) cur[c] = nxt[c];
            }
        }
#pragma unroll
        for (int k = 0; k < IB; k++) { ax[k] += (double)sx[k]; ay[k] += (double)sy[k]; az[k] += (double)sz[k]; }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < IB; k++) {
        const int i = i0 + k * 256;
        if (i < n) { out[3 * i] = ax[k]; out[3 * i + 1] = ay[k]; out[3 * i + 2] = az[k]; }
    }
}

template <int IB, int MODE, int CH>
static int run(const float4 *d, int n, double *o, const char *name, std::vector<double> &ref) {
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const int gb = (n + 256 * IB - 1) / (256 * IB);
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
        HC(hipEventRecord(e0));
        k_direct<IB, MODE, CH><<<gb, 256>>>(d, n, 1.0f, o);
        HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<double> h(3 * 4096);
    HC(hipMemcpy(h.data(), o, h.size() * 8, hipMemcpyDeviceToHost));
    double dev = 0;
    if (ref.empty()) ref = h;
    else for (size_t i = 0; i < h.size(); i++) dev = std::max(dev, fabs(h[i] - ref[i]) / (fabs(ref[i]) + 1e-30));
    const double pairs = (double)n * n;
    printf("%-44s %8.2f ms  %.1f TFLOP/s at 20 flop per pair  %.1f cycles per 64 pairs per SIMD  (max rel diff to the first variant %.1e)\n", name, best,
           20.0 * pairs / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 * 1024.0 / (pairs / 64.0), dev);
    return 0;
}

int main() {
    const int n = 1 << 20;
    std::vector<float4> h(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; i++) {
        float v[3];
        for (int c = 0; c < 3; c++) { s = s * 1664525u + 1013904223u; v[c] = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 600.f; }
        h[i] = make_float4(v[0], v[1], v[2], 0.05f);
    }
    float4 *d; double *o;
    HC(hipMalloc(&d, (size_t)n * 16)); HC(hipMalloc(&o, (size_t)n * 24));
    HC(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    std::vector<double> ref;
    if (run<4, 0, 1>(d, n, o, "library loop, 4 bodies per thread", ref) || run<4, 1, 4>(d, n, o, "4 per thread, chunks of 4 fetched ahead", ref) ||
        run<4, 1, 8>(d, n, o, "4 per thread, chunks of 8 fetched ahead", ref) || run<2, 0, 1>(d, n, o, "library loop, 2 per thread", ref) ||
        run<2, 1, 8>(d, n, o, "2 per thread, chunks of 8 fetched ahead", ref) || run<8, 0, 1>(d, n, o, "library loop, 8 per thread", ref) ||
        run<8, 1, 4>(d, n, o, "8 per thread, chunks of 4 fetched ahead", ref) || run<6, 1, 4>(d, n, o, "6 per thread, chunks of 4 fetched ahead", ref))
        return 1;
    return 0;
}
