// Prices the primitive behind VERDICT r3 item 3 (direct N^2 with the matrix pipe beside the vector pipe) before
// anything is built into the library.
//
// Formulation: v_mfma_f32_4x4x1_16b_f32 = 16 independent 4x4 outer products per instruction, D[b][m][n] += A[b][m] B[b][n],
// lane 4b + n holds column n of block b (4 rows in 4 VGPRs).  With B = the wave's 64 i-bodies (one per lane) and A = four
// j-bodies replicated over the 16 blocks, ONE instruction forms a term of 64 x 4 pair quantities:
//   d2 + eps2 = (|xi|^2 + eps2) + |xj|^2 - 2 xi.xj     4 instructions (C input = |xi|^2 + eps2), coordinates relative to
//                                                      the wave's centre (the 64 i-bodies are key-sorted neighbours)
//   f = rsq(d2)^3                                      3 VALU per register (4 registers = 4 j-bodies)
//   {sum Gm f Xj, sum Gm f}                            4 instructions: A = Gm_j {X, Y, Z, 1}_j (row m = component), B = f
//   a_i = sum Gm f Xj - Xi sum Gm f
// => 8 MFMA + 12 VALU per 256 pairs, against 13 VALU per 64 pairs in the all-VALU kernel.
//   hipcc --offload-arch=gfx950 -O3 -o direct_mfma direct_mfma.hip && ./direct_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// the library's all-VALU kernel (IB = 4), for the A/B on the same box
template <int IB>
__global__ __launch_bounds__(256) void k_valu(const float4 *__restrict__ posm, int n, float eps2, double *__restrict__ out) {
    __shared__ float4 tile[256];
    const int i0 = blockIdx.x * (256 * IB) + threadIdx.x;
    float px[IB], py[IB], pz[IB];
    double ax[IB], ay[IB], az[IB];
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
        for (int k = 0; k < IB; k++) sx[k] = sy[k] = sz[k] = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < 256; jj++) {
            const float4 q = tile[jj];
#pragma unroll
            for (int k = 0; k < IB; k++) {
                const float dx = q.x - px[k], dy = q.y - py[k], dz = q.z - pz[k];
                const float r2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
                const float inv = __builtin_amdgcn_rsqf(r2);
                const float f = q.w * inv * inv * inv;
                sx[k] = fmaf(f, dx, sx[k]); sy[k] = fmaf(f, dy, sy[k]); sz[k] = fmaf(f, dz, sz[k]);
            }
        }
        for (int k = 0; k < IB; k++) { ax[k] += (double)sx[k]; ay[k] += (double)sy[k]; az[k] += (double)sz[k]; }
        __syncthreads();
    }
    for (int k = 0; k < IB; k++) {
        const int i = i0 + k * 256;
        if (i < n) { out[3 * i] = ax[k]; out[3 * i + 1] = ay[k]; out[3 * i + 2] = az[k]; }
    }
}

// 64 i-bodies per wave, 4 waves per block; j-bodies in tiles of 256, re-expressed per wave relative to its centre
template <int UNROLL, int NACC>
__global__ __launch_bounds__(256) void k_mfma(const float4 *__restrict__ posm, int n, float eps2, double *__restrict__ out) {
    __shared__ float4 raw[256];
    __shared__ float4 P1[4][256];  // [wave][j]        {-2X, -2Y, -2Z, |X|^2}
    __shared__ float4 P2[4][256];  // [wave][4 q + c]  Gm_j {X, Y, Z, 1}[c] of the four j of quad q
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float4 pi = i < n ? posm[i] : posm[n - 1];
    const float cx = 0.5f * (wave_min(pi.x) + wave_max(pi.x)), cy = 0.5f * (wave_min(pi.y) + wave_max(pi.y)),
                cz = 0.5f * (wave_min(pi.z) + wave_max(pi.z));
    const float bx = pi.x - cx, by = pi.y - cy, bz = pi.z - cz;
    const float c0 = fmaf(bz, bz, fmaf(by, by, fmaf(bx, bx, eps2)));
    const f4 C0 = {c0, c0, c0, c0};
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (int t = 0; t < (n + 255) / 256; t++) {
        const int j = t * 256 + threadIdx.x;
        raw[threadIdx.x] = j < n ? posm[j] : make_float4(cx, cy, cz, 0.f);
        __syncthreads();
        {
            float gx[4], gy[4], gz[4], gm[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4 q = raw[lane * 4 + r];
                const float X = q.x - cx, Y = q.y - cy, Z = q.z - cz;
                P1[w][lane * 4 + r] = make_float4(-2.f * X, -2.f * Y, -2.f * Z, fmaf(Z, Z, fmaf(Y, Y, X * X)));
                gx[r] = q.w * X; gy[r] = q.w * Y; gz[r] = q.w * Z; gm[r] = q.w;
            }
            P2[w][lane * 4 + 0] = make_float4(gx[0], gx[1], gx[2], gx[3]);
            P2[w][lane * 4 + 1] = make_float4(gy[0], gy[1], gy[2], gy[3]);
            P2[w][lane * 4 + 2] = make_float4(gz[0], gz[1], gz[2], gz[3]);
            P2[w][lane * 4 + 3] = make_float4(gm[0], gm[1], gm[2], gm[3]);
        }
        __syncthreads();
        f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll UNROLL
        for (int q = 0; q < 64; q++) {
            const float4 a = P1[w][q * 4 + (lane & 3)];
            const float4 g = P2[w][q * 4 + (lane & 3)];
            f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, bx, C0, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, by, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, bz, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, 1.0f, d, 0, 0, 0);
            float f0 = __builtin_amdgcn_rsqf(d[0]), f1 = __builtin_amdgcn_rsqf(d[1]), f2 = __builtin_amdgcn_rsqf(d[2]),
                  f3 = __builtin_amdgcn_rsqf(d[3]);
            f0 = f0 * f0 * f0; f1 = f1 * f1 * f1; f2 = f2 * f2 * f2; f3 = f3 * f3 * f3;
            if (NACC == 2 && (q & 1)) {
                acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(g.x, f0, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(g.y, f1, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(g.z, f2, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(g.w, f3, acc2, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(g.x, f0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(g.y, f1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(g.z, f2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(g.w, f3, acc, 0, 0, 0);
            }
        }
        acc += acc2;
        ax += (double)acc[0] - (double)bx * (double)acc[3];
        ay += (double)acc[1] - (double)by * (double)acc[3];
        az += (double)acc[2] - (double)bz * (double)acc[3];
        __syncthreads();
    }
    if (i < n) { out[3 * i] = ax; out[3 * i + 1] = ay; out[3 * i + 2] = az; }
}


// debug: what does the matrix pipe return for d2, against the same fma chain on the vector pipe and against the
// difference form?  One block of 64 i-bodies against the first 256 j-bodies.
__global__ __launch_bounds__(64) void k_probe(const float4 *__restrict__ posm, int n, int i_base, int j_base, float eps2, float *__restrict__ out) {
    __shared__ float4 P1[256];
    const int lane = threadIdx.x;
    const float4 pi = posm[i_base + lane];
    const float cx = 0.5f * (wave_min(pi.x) + wave_max(pi.x)), cy = 0.5f * (wave_min(pi.y) + wave_max(pi.y)),
                cz = 0.5f * (wave_min(pi.z) + wave_max(pi.z));
    const float bx = pi.x - cx, by = pi.y - cy, bz = pi.z - cz;
    const float c0 = fmaf(bz, bz, fmaf(by, by, fmaf(bx, bx, eps2)));
    const f4 C0 = {c0, c0, c0, c0};
    for (int r = 0; r < 4; r++) {
        const float4 q = posm[j_base + lane * 4 + r];
        const float X = q.x - cx, Y = q.y - cy, Z = q.z - cz;
        P1[lane * 4 + r] = make_float4(-2.f * X, -2.f * Y, -2.f * Z, fmaf(Z, Z, fmaf(Y, Y, X * X)));
    }
    __syncthreads();
    float e_chain = 0.f, e_true = 0.f, smax = 0.f;
    for (int q = 0; q < 64; q++) {
        const float4 a = P1[q * 4 + (lane & 3)];
        f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, bx, C0, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, by, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, bz, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, 1.0f, d, 0, 0, 0);
        for (int r = 0; r < 4; r++) {
            const float4 aj = P1[q * 4 + r];
            const float chain = fmaf(aj.w, 1.0f, fmaf(aj.z, bz, fmaf(aj.y, by, fmaf(aj.x, bx, c0))));
            const float4 pj = posm[j_base + q * 4 + r];
            const double dx = (double)pj.x - pi.x, dy = (double)pj.y - pi.y, dz = (double)pj.z - pi.z;
            const double tr = dx * dx + dy * dy + dz * dz + (double)eps2;
            e_chain = fmaxf(e_chain, fabsf(d[r] - chain) / chain);
            e_true = fmaxf(e_true, (float)(fabs((double)d[r] - tr) / tr));
            smax = fmaxf(smax, aj.w);
        }
    }
    out[lane] = e_chain; out[64 + lane] = e_true; out[128 + lane] = smax; out[192 + lane] = fmaf(bz, bz, fmaf(by, by, bx * bx));
}

static unsigned long long morton(float x, float y, float z, float lo, float span) {
    unsigned long long k = 0;
    unsigned xi = (unsigned)((x - lo) / span * 1048575.f), yi = (unsigned)((y - lo) / span * 1048575.f), zi = (unsigned)((z - lo) / span * 1048575.f);
    for (int b = 19; b >= 0; b--) k = (k << 3) | (((xi >> b) & 1) << 2) | (((yi >> b) & 1) << 1) | ((zi >> b) & 1);
    return k;
}

static std::vector<float4> plummer(int n, float a, float rmax, bool disk, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<float4> p(n);
    for (int i = 0; i < n; i++) {
        float r = a / sqrtf(powf(std::max(U(rng), 1e-6f), -2.f / 3.f) - 1.f);
        r = std::min(r, rmax);
        const float ct = 2.f * U(rng) - 1.f, st = sqrtf(1.f - ct * ct), ph = 6.2831853f * U(rng);
        p[i] = make_float4(r * st * cosf(ph), disk ? 0.01f * r * ct : r * ct, r * st * sinf(ph), 0.05f);
    }
    std::vector<std::pair<unsigned long long, int>> key(n);
    for (int i = 0; i < n; i++) key[i] = {morton(p[i].x, p[i].y, p[i].z, -rmax, 2.f * rmax), i};
    std::sort(key.begin(), key.end());
    std::vector<float4> s(n);
    for (int i = 0; i < n; i++) s[i] = p[key[i].second];
    return s;
}

int main(int argc, char **argv) {
    const float eps = argc > 1 ? (float)atof(argv[1]) : 1.0f;
    {
        const int n = 65536;
        std::vector<float4> h = plummer(n, 90.f, 450.f, false, 5);
        float4 *d; float *o;
        HC(hipMalloc(&d, (size_t)n * 16)); HC(hipMalloc(&o, 1024));
        HC(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
        for (int ib : {0, 20000, 32768, 50000}) for (int off : {0, 4096}) {
            const int i_base = ib / 64 * 64, j_base = (i_base + off) % (n - 256);
            k_probe<<<1, 64>>>(d, n, i_base, j_base, eps * eps, o);
            float r[256];
            HC(hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost));
            float ec = 0, et = 0, sm = 0, si = 0;
            for (int l = 0; l < 64; l++) { ec = std::max(ec, r[l]); et = std::max(et, r[64 + l]); sm = std::max(sm, r[128 + l]); si = std::max(si, r[192 + l]); }
            printf("probe i=%6d j=%6d: |mfma - fma chain|/chain max %.2e   |mfma - float64 d2|/d2 max %.2e   max |Xj|^2 %.3g  max |Xi|^2 %.3g\n", i_base, j_base, ec, et, sm, si);
        }
        HC(hipFree(d)); HC(hipFree(o));
    }
    for (int disk = 0; disk < 2; disk++) {
        // accuracy: 16 384 sorted bodies against a float64 all-pairs sum on the host
        for (int n : {16384, 65536}) {
            std::vector<float4> h = plummer(n, 90.f, 450.f, disk, 5);
            float4 *d; double *o;
            HC(hipMalloc(&d, (size_t)n * 16)); HC(hipMalloc(&o, (size_t)n * 24));
            HC(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
            std::vector<double> a1(3 * (size_t)n), a2(3 * (size_t)n);
            k_valu<1><<<(n + 255) / 256, 256>>>(d, n, eps * eps, o);
            HC(hipMemcpy(a1.data(), o, (size_t)n * 24, hipMemcpyDeviceToHost));
            k_mfma<4, 1><<<(n + 255) / 256, 256>>>(d, n, eps * eps, o);
            HC(hipMemcpy(a2.data(), o, (size_t)n * 24, hipMemcpyDeviceToHost));
            const int ns = 1024;
            double e1 = 0, e2 = 0, r1 = 0, r2 = 0;
            for (int s = 0; s < ns; s++) {
                const int i = (int)((long long)s * n / ns);
                double ax = 0, ay = 0, az = 0;
                for (int j = 0; j < n; j++) {
                    const double dx = (double)h[j].x - h[i].x, dy = (double)h[j].y - h[i].y, dz = (double)h[j].z - h[i].z;
                    const double r2_ = dx * dx + dy * dy + dz * dz + (double)eps * eps;
                    const double f = (double)h[j].w / (r2_ * sqrt(r2_));
                    ax += f * dx; ay += f * dy; az += f * dz;
                }
                const double nrm = sqrt(ax * ax + ay * ay + az * az);
                const double d1 = sqrt(pow(a1[3 * i] - ax, 2) + pow(a1[3 * i + 1] - ay, 2) + pow(a1[3 * i + 2] - az, 2)) / nrm;
                const double d2 = sqrt(pow(a2[3 * i] - ax, 2) + pow(a2[3 * i + 1] - ay, 2) + pow(a2[3 * i + 2] - az, 2)) / nrm;
                e1 = std::max(e1, d1); e2 = std::max(e2, d2); r1 += d1 * d1; r2 += d2 * d2;
            }
            printf("%s n=%6d eps=%.2f  relative acceleration error vs float64 (1024 bodies): VALU max %.2e rms %.2e | MFMA max %.2e rms %.2e\n",
                   disk ? "disk   " : "plummer", n, eps, e1, sqrt(r1 / ns), e2, sqrt(r2 / ns));
            HC(hipFree(d)); HC(hipFree(o));
        }
    }
    // rate: 262 144 bodies
    {
        const int n = 262144;
        std::vector<float4> h = plummer(n, 90.f, 450.f, false, 9);
        float4 *d; double *o;
        HC(hipMalloc(&d, (size_t)n * 16)); HC(hipMalloc(&o, (size_t)n * 24));
        HC(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        for (int v = 0; v < 8; v++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                HC(hipEventRecord(e0));
                const int gb = (n + 255) / 256;
                switch (v) {
                case 0: k_valu<4><<<(n + 1023) / 1024, 256>>>(d, n, eps * eps, o); break;
                case 1: k_valu<1><<<gb, 256>>>(d, n, eps * eps, o); break;
                case 2: k_mfma<1, 1><<<gb, 256>>>(d, n, eps * eps, o); break;
                case 3: k_mfma<2, 1><<<gb, 256>>>(d, n, eps * eps, o); break;
                case 4: k_mfma<4, 1><<<gb, 256>>>(d, n, eps * eps, o); break;
                case 5: k_mfma<8, 1><<<gb, 256>>>(d, n, eps * eps, o); break;
                case 6: k_mfma<4, 2><<<gb, 256>>>(d, n, eps * eps, o); break;
                case 7: k_mfma<8, 2><<<gb, 256>>>(d, n, eps * eps, o); break;
                }
                HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
                float ms; HC(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            const double pairs = (double)n * n;
            const char *nm[] = {"VALU IB=4 (1 wave/SIMD at this n)", "VALU IB=1", "MFMA unroll 1", "MFMA unroll 2", "MFMA unroll 4", "MFMA unroll 8",
                                "MFMA unroll 4, 2 accumulators", "MFMA unroll 8, 2 accumulators"};
            printf("%-34s n=%d: %.2f ms  %.3e pairs/s  %.1f TFLOP/s at 20 flop per pair  %.1f cycles per 64 pairs per SIMD\n", nm[v], n, best,
                   pairs / (best * 1e-3), 20.0 * pairs / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 * 1024.0 / (pairs / 64.0));
        }
    }
    return 0;
}
