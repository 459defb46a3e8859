// Micro-benchmark for the "sparse descent" walk variant: what does ONE drain round cost - 64
// lanes = 64 (body, node) pairs popped from an LDS queue, each lane fetching ITS node's 24-byte record with vector
// loads, taking its body's position from LDS, running the fp32 opening test + force, adding the force into the
// body's LDS accumulator, and re-queueing the children of the nodes it opens?  Compared, in the same run, with the
// lock-step visit it would replace (one scalar record fetch, 16 VALU).  The node array is a random permutation
// walk over `span` records (L2-resident like the octree's hot part), pairs of one opened cell are 8 consecutive
// records (its children).  Prints cycles per round / per visit per SIMD at 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct alignas(8) Node { float cx, cy, cz, gm, s2t; unsigned next_off; };
constexpr int kRounds = 512;

// mode 0: ds_add_f32 per lane (8 lanes share a body: same-address adds); mode 1: reduce the 8 lanes of a body with
// DPP-style shuffles, one ds_add per body
template <int MODE>
__global__ __launch_bounds__(256) void k_pair_round(const Node *__restrict__ nodes, unsigned span, float *out, float eps2) {
    __shared__ float4 body[4][64];            // positions of the wave's 64 bodies
    __shared__ float acc[4][64][4];           // their accumulators
    __shared__ uint2 queue[4][1024];          // (body slot, node index) pairs
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    body[w][lane] = make_float4(lane * 0.37f, lane * 0.11f, lane * 0.23f, 1.f);
    acc[w][lane][0] = acc[w][lane][1] = acc[w][lane][2] = 0.f;
    unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    // initial queue: 64 pairs, groups of 8 lanes = one opener body x the 8 children of a cell
    queue[w][lane] = make_uint2((unsigned)(lane >> 3) * 7u % 64u, ((seed >> 8) % (span / 8)) * 8u + (lane & 7));
    unsigned head = 0, tail = 64;
    __syncthreads();
    for (int r = 0; r < kRounds; r++) {
        const uint2 pr = queue[w][(head + lane) & 1023];
        head += 64;
        const float4 p = body[w][pr.x & 63];
        const char *q = reinterpret_cast<const char *>(nodes) + (size_t)(pr.y % span) * 24;
        const float2 a0 = *reinterpret_cast<const float2 *>(q), a1 = *reinterpret_cast<const float2 *>(q + 8);
        const float2 b = *reinterpret_cast<const float2 *>(q + 16);
        const float dx = a0.x - p.x, dy = a0.y - p.y, dz = a1.x - p.z;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
        const bool take = __float_as_int(b.x) < __float_as_int(d2);
        const float inv = __builtin_amdgcn_rsqf(d2);
        const float f = take ? (a1.y * inv) * (inv * inv) : 0.f;
        float fx = dx * f, fy = dy * f, fz = dz * f;
        if (MODE == 0) {
            atomicAdd(&acc[w][pr.x & 63][0], fx);
            atomicAdd(&acc[w][pr.x & 63][1], fy);
            atomicAdd(&acc[w][pr.x & 63][2], fz);
        } else {
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                fx += __shfl_xor(fx, o); fy += __shfl_xor(fy, o); fz += __shfl_xor(fz, o);
            }
            if ((lane & 7) == 0) {
                atomicAdd(&acc[w][pr.x & 63][0], fx);
                atomicAdd(&acc[w][pr.x & 63][1], fy);
                atomicAdd(&acc[w][pr.x & 63][2], fz);
            }
        }
        // openers re-queue their node's children: here a fixed share of the lanes (1 in 8), 8 pairs each, so that the
        // queue keeps 64 pairs per round; the slot arithmetic is the real one (ballot, prefix count)
        seed = seed * 1664525u + 1013904223u;
        const bool opens = (lane & 7) == (r & 7);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(opens);
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (opens) {
            const unsigned base = tail + before * 8u;
            const unsigned child0 = ((__float_as_uint(b.y) / 24u) + (seed >> 12)) % (span / 8) * 8u;
#pragma unroll
            for (int k = 0; k < 8; k++) queue[w][(base + k) & 1023] = make_uint2(pr.x, child0 + k);
        }
        tail += 8u * (unsigned)__builtin_popcountll(m);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[w][lane][0] + acc[w][lane][1] + acc[w][lane][2] + (float)(tail - head);
}

// the lock-step visit it replaces: wave-uniform cursor, scalar record fetch, same arithmetic on all 64 lanes
__global__ __launch_bounds__(256) void k_lockstep(const Node *__restrict__ nodes, unsigned span, float *out, float eps2) {
    const int lane = threadIdx.x & 63;
    const float px = lane * 0.37f, py = lane * 0.11f, pz = lane * 0.23f;
    float ax = 0.f, ay = 0.f, az = 0.f;
    unsigned cur = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u % span;
    for (int r = 0; r < kRounds * 8; r++) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        const Node nd = nodes[cur];
        const float dx = nd.cx - px, dy = nd.cy - py, dz = nd.cz - pz;
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
        const bool take = __float_as_int(nd.s2t) < __float_as_int(d2);
        const float inv = __builtin_amdgcn_rsqf(d2);
        const float f = take ? (nd.gm * inv) * (inv * inv) : 0.f;
        ax = fmaf(dx, f, ax); ay = fmaf(dy, f, ay); az = fmaf(dz, f, az);
        cur = (__builtin_amdgcn_ballot_w64(!take) ? cur + 1 : nd.next_off / 24u) % span;
    }
    out[blockIdx.x * 256 + threadIdx.x] = ax + ay + az;
}

int main() {
    const unsigned span = 1u << 17;  // 128 k records = 3 MB: L2-resident, as the hot part of the octree is
    std::vector<Node> h(span);
    srand(1);
    for (unsigned i = 0; i < span; i++) {
        h[i] = Node{(float)(rand() % 1000) * 0.1f, (float)(rand() % 1000) * 0.1f, (float)(rand() % 1000) * 0.1f, 1.f,
                    (rand() % 4) ? 10.f : 1e9f, (unsigned)((i + 1 + rand() % 64) % span) * 24u};
    }
    Node *d; float *out;
    HC(hipMalloc(&d, span * sizeof(Node)));
    HC(hipMemcpy(d, h.data(), span * sizeof(Node), hipMemcpyHostToDevice));
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    HC(hipMalloc(&out, blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    float ms;
    const double wave_rounds_per_simd = (double)blocks * 4 / 1024.0 * kRounds;
    for (int rep = 0; rep < 2; rep++) {
        HC(hipEventRecord(e0)); k_pair_round<0><<<blocks, 256>>>(d, span, out, 2.25f); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("pair round, ds_add per lane:            %.3f ms  %.0f cycles per round per SIMD (64 pairs) @2.4 GHz\n", ms, ms * 1e-3 * 2.4e9 / wave_rounds_per_simd);
        HC(hipEventRecord(e0)); k_pair_round<1><<<blocks, 256>>>(d, span, out, 2.25f); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("pair round, 8-lane reduce + ds_add:     %.3f ms  %.0f cycles per round per SIMD (64 pairs) @2.4 GHz\n", ms, ms * 1e-3 * 2.4e9 / wave_rounds_per_simd);
        HC(hipEventRecord(e0)); k_lockstep<<<blocks, 256>>>(d, span, out, 2.25f); HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
        HC(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("lock-step visit (C++ form, 8 per round): %.3f ms  %.0f cycles per visit per SIMD @2.4 GHz\n", ms, ms * 1e-3 * 2.4e9 / (wave_rounds_per_simd * 8));
    }
    return 0;
}
