// Counter-based dense sketch generation on the GPU (Philox4x32-10 -> N(0,1) or U(-1,1)).
//
// Stands in for RandBLAS::fill_dense / RandBLAS::RNGState at the reference's call sites
// (RandLAPACK/comps/rl_rs.hh:134-139, drivers/rl_bqrrp.hh:310-311, drivers/rl_hqrrp.hh:929-930,
// drivers/rl_abrik.hh:298-299, drivers/rl_cqrrpt.hh:351-352).  RandBLAS itself is an un-vendored
// submodule (pinned 04f2018a...) and no reference test pins a sketch entry, so the stream below is this
// library's own, fully specified here ("parity unpinned" for the random stream):
//
//   state  = (ctr[4], key[2])  32-bit words, ctr is a 128-bit little-endian counter
//   block b (b = 0,1,...) : r[0..3] = Philox4x32-10(ctr + b, key)
//   Gaussian : u0=(r0+0.5)/2^32, u1=(r1+0.5)/2^32, rad=sqrt(-2 ln u1):  z0=rad*cos(2 pi u0), z1=rad*sin(2 pi u0)
//              same with (r2,r3) -> z2,z3
//   Uniform  : z_e = (r_e + 0.5) / 2^31 - 1                      in (-1,1)
//   entries 4b..4b+3 of the rows x cols buffer in COLUMN-MAJOR linear order (ld = rows, as the reference
//   reads every fill_dense buffer) receive z0..z3;
//   next state: ctr + ceil(rows*cols/4).
// All floating point is done in fp64 and rounded once to T, so the fp32 stream is the rounded fp64 one.
// Because any block can be regenerated from (ctr,key) alone, every GPU of a row-sharded run can produce
// exactly the slice of the sketch it needs with no communication.
#include "rlhip_internal.h"

namespace {

__host__ __device__ inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ inline void ctr_add(const uint32_t base[4], uint64_t inc, uint32_t out[4]) {
    uint64_t lo = ((uint64_t)base[1] << 32) | base[0];
    uint64_t hi = ((uint64_t)base[3] << 32) | base[2];
    uint64_t nlo = lo + inc;
    if (nlo < lo) hi += 1;
    out[0] = (uint32_t)nlo; out[1] = (uint32_t)(nlo >> 32);
    out[2] = (uint32_t)hi;  out[3] = (uint32_t)(hi >> 32);
}

struct RngState {
    uint32_t ctr[4];
    uint32_t key[2];
};

template <typename T>
__global__ void fill_dense_kernel(int dist, int64_t total, T* __restrict__ buf, RngState st) {
    int64_t nblk = (total + 3) / 4;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
         b += (int64_t)gridDim.x * blockDim.x) {
        uint32_t c[4], r[4];
        ctr_add(st.ctr, (uint64_t)b, c);
        philox4x32_10(c, st.key, r);
        double z[4];
        if (dist == 0) {
            const double s32 = 2.3283064365386963e-10;  // 2^-32
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double u0 = ((double)r[2 * h] + 0.5) * s32;
                double u1 = ((double)r[2 * h + 1] + 0.5) * s32;
                double rad = sqrt(-2.0 * log(u1));
                double sn, cs;
                sincospi(2.0 * u0, &sn, &cs);
                z[2 * h] = rad * cs;
                z[2 * h + 1] = rad * sn;
            }
        } else {
            const double s31 = 4.6566128730773926e-10;  // 2^-31
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = ((double)r[e] + 0.5) * s31 - 1.0;
        }
        int64_t base = 4 * b;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (base + e < total) buf[base + e] = (T)z[e];
    }
}

// Rows [row0, row0 + loc_rows) of the glob_rows x cols matrix that fill_dense_kernel would produce: element (i, j) of
// the global matrix is stream position i + j*glob_rows, so a row shard regenerates exactly its slice of the global
// operator (the device counterpart of RandBLAS::fill_dense_unpacked's row/column offsets).
template <typename T>
__global__ void fill_dense_rows_kernel(int dist, int64_t glob_rows, int64_t cols, int64_t row0, int64_t loc_rows, T* __restrict__ buf,
                                       int64_t ld, RngState st) {
    const int64_t total = loc_rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e % loc_rows, j = e / loc_rows;
        const int64_t g = (row0 + i) + j * glob_rows;
        uint32_t c[4], r[4];
        ctr_add(st.ctr, (uint64_t)(g >> 2), c);
        philox4x32_10(c, st.key, r);
        const int w = (int)(g & 3);
        double z;
        if (dist == 0) {
            const double s32 = 2.3283064365386963e-10;
            const int h = w >> 1;
            double u0 = ((double)r[2 * h] + 0.5) * s32;
            double u1 = ((double)r[2 * h + 1] + 0.5) * s32;
            double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincospi(2.0 * u0, &sn, &cs);
            z = (w & 1) ? rad * sn : rad * cs;
        } else {
            z = ((double)r[w] + 0.5) * 4.6566128730773926e-10 - 1.0;
        }
        buf[i + j * ld] = (T)z;
    }
}

__global__ void philox_raw_kernel(int64_t nblk, uint32_t* __restrict__ out, RngState st) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    uint32_t c[4], r[4];
    ctr_add(st.ctr, (uint64_t)b, c);
    philox4x32_10(c, st.key, r);
    for (int e = 0; e < 4; ++e) out[4 * b + e] = r[e];
}

}  // namespace

namespace rlhip {

template <typename T>
int fill_dense(rlhip_ctx* c, int dist, int64_t rows, int64_t cols, T* buf, const uint32_t ctr[4],
               const uint32_t key[2], uint32_t next_ctr[4]) {
    if (rows < 0) return -3;
    if (cols < 0) return -4;
    if (dist != 0 && dist != 1) return -2;
    int64_t total = rows * cols;
    int64_t nblk = (total + 3) / 4;
    RngState st;
    for (int i = 0; i < 4; ++i) st.ctr[i] = ctr[i];
    st.key[0] = key[0]; st.key[1] = key[1];
    if (next_ctr) ctr_add(ctr, (uint64_t)nblk, next_ctr);
    if (total == 0) return 0;
    int64_t blocks = (nblk + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fill_dense_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, c->stream, dist, total, buf, st);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template <typename T>
int fill_dense_rows(rlhip_ctx* c, int dist, int64_t glob_rows, int64_t cols, int64_t row0, int64_t loc_rows, T* buf, int64_t ld,
                    const uint32_t ctr[4], const uint32_t key[2], uint32_t next_ctr[4]) {
    if (dist != 0 && dist != 1) return -2;
    if (glob_rows < 0) return -3;
    if (cols < 0) return -4;
    if (row0 < 0 || loc_rows < 0 || row0 + loc_rows > glob_rows) return -5;
    if (ld < (loc_rows > 1 ? loc_rows : 1)) return -8;
    RngState st;
    for (int i = 0; i < 4; ++i) st.ctr[i] = ctr[i];
    st.key[0] = key[0]; st.key[1] = key[1];
    if (next_ctr) ctr_add(ctr, (uint64_t)((glob_rows * cols + 3) / 4), next_ctr);
    const int64_t total = loc_rows * cols;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fill_dense_rows_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, c->stream, dist, glob_rows, cols, row0, loc_rows, buf, ld,
                       st);
    RLHIP_LAUNCH_CHECK();
    return 0;
}
template int fill_dense_rows<double>(rlhip_ctx*, int, int64_t, int64_t, int64_t, int64_t, double*, int64_t, const uint32_t*, const uint32_t*, uint32_t*);
template int fill_dense_rows<float>(rlhip_ctx*, int, int64_t, int64_t, int64_t, int64_t, float*, int64_t, const uint32_t*, const uint32_t*, uint32_t*);

int philox_raw(rlhip_ctx* c, int64_t nblk, uint32_t* out_dev, const uint32_t ctr[4], const uint32_t key[2]) {
    if (nblk <= 0) return 0;
    RngState st;
    for (int i = 0; i < 4; ++i) st.ctr[i] = ctr[i];
    st.key[0] = key[0]; st.key[1] = key[1];
    hipLaunchKernelGGL(philox_raw_kernel, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, c->stream, nblk,
                       out_dev, st);
    RLHIP_LAUNCH_CHECK();
    return 0;
}

template int fill_dense<double>(rlhip_ctx*, int, int64_t, int64_t, double*, const uint32_t*, const uint32_t*, uint32_t*);
template int fill_dense<float>(rlhip_ctx*, int, int64_t, int64_t, float*, const uint32_t*, const uint32_t*, uint32_t*);

}  // namespace rlhip
