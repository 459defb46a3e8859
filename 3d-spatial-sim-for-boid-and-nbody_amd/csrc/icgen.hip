// Device-side initial conditions: the three distributions BASELINE.json names,
// drawn on the GPU so that a 10 M-body start does not wait for seconds of host NumPy and a 640 MB
// upload.  Same formulas as the reference's generate_distribution (tools/presets.py:104-232 galaxy /
// collision via the disk helper and compute_rotation_curve :52-88, :350-397 cluster); the random
// stream is a counter-based Philox4x32-10 (key = seed, counter = body index, draw block), so the
// result depends only on (seed, n, R, G) - STATISTICAL parity with the NumPy generator, not bit
// parity (tests/test_gpu_icgen.py compares the distributions).
#include <cstring>
#include <math.h>

#include "common.h"

namespace nbmi {

namespace {

constexpr int kBlock = 256;
constexpr double kTwoPi = 6.283185307179586476925286766559;

struct Draws {
    uint2 key;
    uint32_t body_lo, body_hi;
    // uniform double in (0,1) number `k` of this body (two 32-bit words of Philox block k/2)
    __device__ double uniform(uint32_t k) const {
        uint32_t ctr[4] = {body_lo, body_hi, k >> 1, 0x49436e62u};  // last word: stream tag
        uint32_t out[4];
        const uint32_t key2[2] = {key.x, key.y};
        philox4x32_10(ctr, key2, out);
        const uint32_t a = (k & 1) ? out[2] : out[0], b = (k & 1) ? out[3] : out[1];
        const uint64_t bits = ((uint64_t)a << 32 | b) >> 11;  // 53 bits
        return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
    }
    // standard normal pair number `k` (Box-Muller on uniforms 2k, 2k+1)
    __device__ void normal2(uint32_t k, double &n0, double &n1) const {
        const double u1 = uniform(2 * k), u2 = uniform(2 * k + 1);
        const double rad = sqrt(-2.0 * log(u1));
        double s, c;
        sincos(kTwoPi * u2, &s, &c);
        n0 = rad * c;
        n1 = rad * s;
    }
};

__device__ inline Draws draws_for(uint64_t seed, int64_t body) {
    Draws d;
    d.key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    d.body_lo = (uint32_t)body;
    d.body_hi = (uint32_t)((uint64_t)body >> 32);
    return d;
}

struct DiskP {
    double R, G, scale_length, softening, max_r, height, disp, spin, x0, y0, vx_add;
    int64_t first, count;  // bodies [first, first + count) form this disk
};

// positions of one exponential disk + the radius sort key (presets.py:110-127 / :161-178)
__global__ __launch_bounds__(kBlock) void k_disk_positions(DiskP P, uint64_t seed, double *__restrict__ x,
                                                           double *__restrict__ y, double *__restrict__ z,
                                                           double *__restrict__ radius, double *__restrict__ angle,
                                                           uint64_t *__restrict__ rkey, uint32_t *__restrict__ ridx) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= P.count) return;
    const int64_t i = P.first + k;
    const Draws d = draws_for(seed, i);
    double r = -P.scale_length * log(d.uniform(0));          // exponential(scale_length)
    r = r * (1.0 - exp(-P.max_r / (r + 0.01)));                // soft truncation
    r = fmax(r, P.R * 0.001);
    const double th = kTwoPi * d.uniform(1);
    double n0, n1;
    d.normal2(1, n0, n1);                                      // uniforms 2,3
    const double disk_height = P.R * P.height * (1.0 + sqrt(r / P.R) * 0.3);
    double s, c;
    sincos(th, &s, &c);
    x[i] = r * c + P.x0;
    y[i] = n0 * disk_height + P.y0;
    z[i] = r * s;
    radius[i] = r;
    angle[i] = th;
    rkey[k] = (uint64_t)__double_as_longlong(r);               // positive doubles order like their bits
    ridx[k] = (uint32_t)k;
}

// velocities from the rotation curve: enclosed mass = (rank of the radius + 1) unit masses
// (compute_rotation_curve, presets.py:52-88) + dispersion (:136-146)
__global__ __launch_bounds__(kBlock) void k_disk_velocities(DiskP P, uint64_t seed, const double *__restrict__ radius,
                                                            const double *__restrict__ angle,
                                                            const uint32_t *__restrict__ sorted_idx,
                                                            double *__restrict__ vx, double *__restrict__ vy,
                                                            double *__restrict__ vz) {
    const int64_t rank = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (rank >= P.count) return;
    const int64_t i = P.first + sorted_idx[rank];
    const Draws d = draws_for(seed, i);
    const double r = radius[i], th = angle[i];
    const double enclosed = (double)(rank + 1);
    const double eps = P.softening * 2.0, eps_sq = eps * eps, r_sq = r * r;
    double speed = sqrt(P.G * enclosed * r_sq / pow(r_sq + eps_sq, 1.5));
    speed *= fmax(r_sq / (r_sq + eps_sq), 0.3);                // inner_scale = softening*2 = eps
    double s, c;
    sincos(th, &s, &c);
    double ux = -speed * s * P.spin, uz = speed * c * P.spin;
    const double radial_factor = r / (r + P.softening * 2.0);
    const double sigma = speed * P.disp * radial_factor + sqrt(P.G * (double)P.count * 0.00005);
    double n0, n1, n2, n3;
    d.normal2(2, n0, n1);                                      // uniforms 4..7
    d.normal2(3, n2, n3);
    vx[i] = ux + n0 * sigma + P.vx_add;
    vz[i] = uz + n1 * sigma;
    vy[i] = n2 * sigma * 0.25;
}

// Plummer sphere (presets.py:350-397)
__global__ __launch_bounds__(kBlock) void k_cluster(int64_t n, double R, double G, uint64_t seed, double *__restrict__ x,
                                                    double *__restrict__ y, double *__restrict__ z,
                                                    double *__restrict__ vx, double *__restrict__ vy,
                                                    double *__restrict__ vz) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Draws d = draws_for(seed, i);
    const double a = R * 0.3;
    const double u = d.uniform(0);
    double r = a / sqrt(pow(u, -2.0 / 3.0) - 1.0);
    r = fmin(fmax(r, 0.0), R * 1.5);
    const double phi = kTwoPi * d.uniform(1);
    const double cos_t = 2.0 * d.uniform(2) - 1.0;
    const double sin_t = sqrt(1.0 - cos_t * cos_t);
    double s, c;
    sincos(phi, &s, &c);
    x[i] = r * sin_t * c;
    y[i] = r * cos_t;
    z[i] = r * sin_t * s;
    const double total_mass = (double)n * 0.001;
    const double ra = r / a;
    const double base = G * total_mass / (6.0 * a);
    const double sigma = sqrt(fmax(base / sqrt(1.0 + ra * ra), base * 0.01));
    double n0, n1;
    d.normal2(2, n0, n1);                                      // uniforms 4,5
    const double v_mag = fabs(n0 * sigma * sqrt(3.0));
    const double v_phi = kTwoPi * d.uniform(6);
    const double v_cos = 2.0 * d.uniform(7) - 1.0;
    const double v_sin = sqrt(1.0 - v_cos * v_cos);
    sincos(v_phi, &s, &c);
    vx[i] = v_mag * v_sin * c;
    vy[i] = v_mag * v_cos;
    vz[i] = v_mag * v_sin * s;
}

__global__ __launch_bounds__(kBlock) void k_fill_mass_id(int64_t n, double *__restrict__ m, int32_t *__restrict__ id) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    m[i] = 1.0;
    id[i] = (int32_t)i;
}

// centre-of-mass velocity (unit masses): fixed grid + fixed order => deterministic
constexpr int kSumBlocks = 256;
__global__ __launch_bounds__(kBlock) void k_sum3(const double *__restrict__ a, const double *__restrict__ b,
                                                 const double *__restrict__ c, int64_t n, double *__restrict__ part) {
    __shared__ double red[3][kBlock];
    double s0 = 0, s1 = 0, s2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        s0 += a[i]; s1 += b[i]; s2 += c[i];
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; k++) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; k++) part[3 * blockIdx.x + k] = red[k][0];
}

__global__ __launch_bounds__(kBlock) void k_sub_mean3(double *__restrict__ a, double *__restrict__ b,
                                                      double *__restrict__ c, int64_t n, const double *__restrict__ part,
                                                      int nparts) {
    double m0 = 0, m1 = 0, m2 = 0;
    for (int p = 0; p < nparts; p++) { m0 += part[3 * p]; m1 += part[3 * p + 1]; m2 += part[3 * p + 2]; }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    a[i] -= m0 / (double)n;
    b[i] -= m1 / (double)n;
    c[i] -= m2 / (double)n;
}

inline int nblocks(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

struct Scratch {
    double *radius = nullptr, *angle = nullptr, *part = nullptr;
    uint64_t *rkey = nullptr, *rkey_s = nullptr;
    uint32_t *ridx = nullptr, *ridx_s = nullptr;
    void *tmp = nullptr;
    ~Scratch() {
        for (void *p : {(void *)radius, (void *)angle, (void *)part, (void *)rkey, (void *)rkey_s, (void *)ridx,
                        (void *)ridx_s, tmp})
            if (p) (void)hipFree(p);
    }
};

int disk(const DiskP &P, uint64_t seed, IcArrays A, Scratch &S, size_t tmp_bytes, hipStream_t st) {
    if (P.count == 0) return 0;
    k_disk_positions<<<nblocks(P.count), kBlock, 0, st>>>(P, seed, A.x, A.y, A.z, S.radius, S.angle, S.rkey, S.ridx);
    NBMI_HIP_CHECK(sort_pairs_u64_u32(S.tmp, tmp_bytes, S.rkey, S.rkey_s, S.ridx, S.ridx_s, (size_t)P.count, 0, 63, st));
    k_disk_velocities<<<nblocks(P.count), kBlock, 0, st>>>(P, seed, S.radius, S.angle, S.ridx_s, A.vx, A.vy, A.vz);
    NBMI_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

int ic_generate(int distribution, int64_t n, double R, double G, uint64_t seed, IcArrays A, hipStream_t st) {
    if (n == 0) return 0;
    Scratch S;
    const size_t tmp_bytes = sort_pairs_temp_bytes((size_t)n, 0, 63);
    NBMI_HIP_CHECK(hipMalloc((void **)&S.radius, n * 8));
    NBMI_HIP_CHECK(hipMalloc((void **)&S.angle, n * 8));
    NBMI_HIP_CHECK(hipMalloc((void **)&S.part, kSumBlocks * 3 * 8));
    NBMI_HIP_CHECK(hipMalloc((void **)&S.rkey, n * 8));
    NBMI_HIP_CHECK(hipMalloc((void **)&S.rkey_s, n * 8));
    NBMI_HIP_CHECK(hipMalloc((void **)&S.ridx, n * 4));
    NBMI_HIP_CHECK(hipMalloc((void **)&S.ridx_s, n * 4));
    NBMI_HIP_CHECK(hipMalloc(&S.tmp, tmp_bytes + 256));
    NBMI_HIP_CHECK(sort_init_temp(S.tmp, st));
    k_fill_mass_id<<<nblocks(n), kBlock, 0, st>>>(n, A.m, A.id);
    bool remove_com = false;
    if (distribution == NBMI_IC_GALAXY) {  // presets.py:104-146
        DiskP P{R, G, R * 0.3, R * 0.03, R * 1.0, 0.012, 0.12, +1.0, 0.0, 0.0, 0.0, 0, n};
        if (int rc = disk(P, seed, A, S, tmp_bytes, st)) return rc;
        remove_com = true;
    } else if (distribution == NBMI_IC_COLLISION) {  // presets.py:148-232
        const int64_t half = n / 2;
        const double separation = (R * 0.5) * 3.5;
        const double total_mass = (double)n * 0.001;
        const double collision_speed = sqrt(2.0 * G * total_mass / separation) * 0.6;
        DiskP P1{R, G, R * 0.25, R * 0.025, R * 0.5, 0.01, 0.10, +1.0, -separation / 2, 0.0, +collision_speed, 0, half};
        DiskP P2{R, G, R * 0.25, R * 0.025, R * 0.5, 0.01, 0.10, -1.0, +separation / 2, R * 0.15, -collision_speed, half,
                 n - half};
        if (int rc = disk(P1, seed, A, S, tmp_bytes, st)) return rc;
        if (int rc = disk(P2, seed, A, S, tmp_bytes, st)) return rc;
    } else if (distribution == NBMI_IC_CLUSTER) {  // presets.py:350-397
        k_cluster<<<nblocks(n), kBlock, 0, st>>>(n, R, G, seed, A.x, A.y, A.z, A.vx, A.vy, A.vz);
        remove_com = true;
    } else {
        set_error("unknown distribution %d", distribution);
        return -1;
    }
    if (remove_com) {
        k_sum3<<<kSumBlocks, kBlock, 0, st>>>(A.vx, A.vy, A.vz, n, S.part);
        k_sub_mean3<<<nblocks(n), kBlock, 0, st>>>(A.vx, A.vy, A.vz, n, S.part, kSumBlocks);
    }
    NBMI_HIP_CHECK(hipGetLastError());
    unsigned sort_err = 0u;
    NBMI_HIP_CHECK(sort_error_word(S.tmp, &sort_err, st));
    NBMI_HIP_CHECK(hipStreamSynchronize(st));  // scratch is freed on return
    if (sort_err) {
        set_error("device radix sort: a look-back spin timed out while ranking the generated bodies");
        return -2;
    }
    return 0;
}

}  // namespace nbmi
