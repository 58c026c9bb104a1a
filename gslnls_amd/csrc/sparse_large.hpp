// sparse_large.hpp -- gsl_nls_large with a SPARSE Jacobian supplied by the caller
// (dgCMatrix / dgRMatrix / dgTMatrix inputs of the reference, src/nls_large.c:16-49, :575-622).
//
// In the reference the model and its Jacobian are R closures; every product the trust-region / CG loop
// needs (J u, J^T u, J^T J) goes through gsl_df_large, which RE-EVALUATES the closure and then runs a GSL
// spblas product on the host.  Here the closures stay what they are -- host callbacks,
// called once per trial point for f and once for J -- and everything n- or nnz-sized after that lives on
// the device: residual weighting, ssr, g = J^T f, diag(J^T J), the fused CG product J^T (J u) as two
// SpMVs, the dense J^T J for the `lm` variant and the covariance.
//
// Layout.  Any input format is brought to one canonical CSR (rows sorted, columns sorted inside a row,
// duplicate triplets summed) plus its transpose index: CSC column pointers, row indices and a permutation
// `perm` into the CSR value array; after every Jacobian evaluation one gather lays the values out in CSC
// order as well, so that both products stream contiguous (value, index) arrays.  The pattern is built on the host when it first appears (or
// changes); afterwards an evaluation uploads nnz values.
//
// Kernels are gather-only and fixed-order (no atomics): results are bit-identical run to run.
//   segment kernels: L = 1 / 4 / 16 / 64 lanes per row (column), chosen from the mean segment length so
//                    that the strided reads stay coalesced; DPP sums inside the group; long segments
//                    (> SP_LONG entries) of a short-segment matrix go to the 64-lane kernel through a list.
// All are HBM-bound: 12 B per stored entry (value + index) + the gathered vector (L2-resident for p << nnz).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include "../../include/gslnls_core.h"
#include "large_host.hpp"

#include "sparse_cg.hpp"

namespace gslnls
{

constexpr int SP_T = 256;
constexpr int SP_LONG = 256; // segments longer than this go to the wavefront kernel in scalar mode

// sum over a group of L consecutive lanes (L = 4, 16: DPP inside a row of 16 lanes; L = 64: whole wavefront),
// fixed order, result valid in the group's first lane
template <int L>
__device__ __forceinline__ double group_sum(double v)
{
    if (L == 64)
        return wave_sum(v);
    v += dpp_mov<0xB1>(v); // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v); // quad_perm [2,3,0,1]
    if (L == 16)
    {
        v += dpp_mov<0x141>(v); // row_half_mirror
        v += dpp_mov<0x140>(v); // row_mirror
    }
    return v;
}

// out[s] = sum_k val[k] * vec[idx[k]] over segment s = [ptr[s], ptr[s+1]); optionally sq[s] = sum val^2.
// L lanes work on one segment (L = 1: a thread per segment; 4 / 16 / 64: coalesced strides + DPP sum), chosen
// from the mean segment length; `list` (optional) names the segments to process -- the long ones of an
// otherwise short-segment matrix (skip_long = they are left to that second launch).
template <int L>
__global__ __launch_bounds__(SP_T) void sp_segment_kernel(const int *__restrict__ ptr, const int *__restrict__ idx,
                                                          const double *__restrict__ val, const double *__restrict__ vec,
                                                          int nseg, const int *__restrict__ list, int nlist, int skip_long,
                                                          double *out, double *sq, const int *flag, int run_when)
{
    // device-resident CG (sparse_cg.hpp): kernels of iterations enqueued past the end of the step do nothing
    if (flag && *flag != run_when)
        return;
    const long gid = (long)blockIdx.x * SP_T + threadIdx.x;
    const long w = gid / L;
    const int lane = (int)(gid % L);
    const bool live = w < (list ? nlist : nseg);
    const int s = live ? (list ? list[w] : (int)w) : 0;
    int b = 0, e = 0;
    if (live)
    {
        b = ptr[s];
        e = ptr[s + 1];
        if (skip_long && e - b > SP_LONG)
            e = b;
    }
    const bool skipped = live && skip_long && ptr[s + 1] - ptr[s] > SP_LONG;
    double a = 0.0, q = 0.0;
    for (int k = b + lane; k < e; k += L)
    {
        const double v = val[k];
        a = fma(v, vec[idx[k]], a);
        q = fma(v, v, q);
    }
    if (L > 1)
    {
        a = group_sum<L>(a);
        q = group_sum<L>(q);
    }
    if (live && !skipped && lane == 0)
    {
        out[s] = a;
        if (sq)
            sq[s] = q;
    }
}

// ---- the dense operator (GSLNLS_SPARSE_DENSE, round 5): J as an n x p block, column-major -------------------------------
// out[i] = sum_k J[i + n k] vec[k]: a lane per row, coalesced over the rows of every column (dgemv, src/nls_large.c:629)
static __global__ __launch_bounds__(SP_T) void dn_rows_kernel(const double *__restrict__ J, const double *__restrict__ vec, int n, int p,
                                                              double *out, const int *flag, int run_when)
{
    if (flag && *flag != run_when)
        return;
    __shared__ double v_s[1024];
    const int i = blockIdx.x * SP_T + threadIdx.x;
    double a = 0.0;
    for (int k0 = 0; k0 < p; k0 += 1024)
    {
        const int kn = p - k0 < 1024 ? p - k0 : 1024;
        __syncthreads();
        for (int k = threadIdx.x; k < kn; k += SP_T)
            v_s[k] = vec[k0 + k];
        __syncthreads();
        if (i < n)
            for (int k = 0; k < kn; ++k)
                a = fma(J[(size_t)(k0 + k) * n + i], v_s[k], a);
    }
    if (i < n)
        out[i] = a;
}
// out[c] = sum_i J[i + n c] vec[i], sq[c] = sum_i J[i + n c]^2: a workgroup per column (contiguous in memory)
static __global__ __launch_bounds__(SP_T) void dn_cols_kernel(const double *__restrict__ J, const double *__restrict__ vec, int n, double *out,
                                                              double *sq, const int *flag, int run_when)
{
    if (flag && *flag != run_when)
        return;
    __shared__ double lds[2 * (SP_T / 64)];
    const double *col = J + (size_t)blockIdx.x * n;
    double a = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += SP_T)
    {
        const double v = col[i];
        a = fma(v, vec[i], a);
        q = fma(v, v, q);
    }
    a = group_sum<64>(a);
    q = group_sum<64>(q);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
    {
        lds[wave] = a;
        lds[SP_T / 64 + wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double ta = 0.0, tq = 0.0;
        for (int w = 0; w < SP_T / 64; ++w)
        {
            ta += lds[w];
            tq += lds[SP_T / 64 + w];
        }
        out[blockIdx.x] = ta;
        if (sq)
            sq[blockIdx.x] = tq;
    }
}
// J^T J of a dense n x p block on the matrix cores (bd_syrk_kernel + its reduction, bd_models.hip); *cpart: scratch kept by the caller
int bd_dense_jtj(const double *d_J, int n, int p, double *d_C, hipStream_t st, double **cpart, size_t *cpart_bytes);

// values in CSC order (one gather per Jacobian evaluation, so that both products stream contiguous arrays)
__global__ __launch_bounds__(SP_T) void sp_permute_kernel(const double *val, const int *perm, long nnz, double *valc)
{
    const long k = (long)blockIdx.x * SP_T + threadIdx.x;
    if (k < nnz)
        valc[k] = val[perm[k]];
}

// f <- (finite(m) ? m - y : +Inf) * sw   (gsl_f_large src/nls_large.c:426-472 + weighting), block partials of f^2
__global__ __launch_bounds__(SP_T) void sp_resid_kernel(double *f, const double *y, const double *sw, int n, double *partial)
{
    __shared__ double lds[SP_T / 64];
    double a = 0.0;
    for (int i = blockIdx.x * SP_T + threadIdx.x; i < n; i += gridDim.x * SP_T)
    {
        const double m = f[i];
        double r = isfinite(m) ? m - y[i] : INFINITY;
        if (sw)
            r *= sw[i];
        f[i] = r;
        a = fma(r, r, a);
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0)
        lds[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = 0.0;
        for (int w = 0; w < SP_T / 64; ++w)
            t += lds[w];
        partial[blockIdx.x] = t;
    }
}

// block partials of w^2
__global__ __launch_bounds__(SP_T) void sp_sumsq_kernel(const double *w, int n, double *partial, const int *flag, int run_when)
{
    __shared__ double lds[SP_T / 64];
    if (flag && *flag != run_when)
        return;
    double a = 0.0;
    for (int i = blockIdx.x * SP_T + threadIdx.x; i < n; i += gridDim.x * SP_T)
        a = fma(w[i], w[i], a);
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0)
        lds[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double t = 0.0;
        for (int k = 0; k < SP_T / 64; ++k)
            t += lds[k];
        partial[blockIdx.x] = t;
    }
}

// dense J^T J (row j of the output owned by one wavefront): for every entry (i, j) of column j, in order,
// lanes walk row i and add v_ij * v_ik into out[j][k] -- columns k are distinct inside a row, so the
// read-modify-writes of one step never collide; steps are sequential.  (reference: gsl_spblas_dgemm + sp2d,
// src/nls_large.c:639-647)
__global__ __launch_bounds__(SP_T) void sp_jtj_kernel(const int *colptr, const int *rowidx, const int *perm,
                                                      const int *rowptr, const int *colidx, const double *val, int p,
                                                      double *out)
{
    const int j = (blockIdx.x * SP_T + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (j >= p)
        return;
    double *row = out + (size_t)j * p;
    for (int c = colptr[j]; c < colptr[j + 1]; ++c)
    {
        const int i = rowidx[c];
        const double vij = val[perm[c]];
        for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64)
            row[colidx[k]] = fma(vij, val[k], row[colidx[k]]);
    }
}

struct SparsePattern
{
    int n = 0, p = 0;
    long nnz = 0;                        // canonical entries (duplicates merged)
    std::vector<int> rowptr, colidx;     // CSR
    std::vector<int> colptr, rowidx, perm; // CSC view: perm[c] = CSR position of CSC entry c
    std::vector<int> map;                // input entry e -> CSR position
    std::vector<int> long_rows, long_cols;
    // fingerprint of the input pattern it was built from
    int in_format = -1;
    long in_nnz = -1;
    std::vector<int> in_a, in_b;
};

// canonical CSR from a dgR (rows: p = row pointers, j = columns), dgC (p = column pointers, i = rows) or
// dgT (i, j triplets, duplicates summed like Matrix does) description
inline int sparse_build_pattern(const gslnls_sparse &J, int n, int p, SparsePattern &P)
{
    if (J.nrow != n || J.ncol != p || J.nnz < 0)
        return GSLNLS_EINVAL;
    const long m = J.nnz;
    std::vector<int> ri(m), ci(m);
    if (J.format == GSLNLS_SPARSE_CSR)
    {
        for (int r = 0; r < n; ++r)
            for (int k = J.p[r]; k < J.p[r + 1]; ++k)
            {
                ri[k] = r;
                ci[k] = J.j[k];
            }
    }
    else if (J.format == GSLNLS_SPARSE_CSC)
    {
        for (int c = 0; c < p; ++c)
            for (int k = J.p[c]; k < J.p[c + 1]; ++k)
            {
                ri[k] = J.i[k];
                ci[k] = c;
            }
    }
    else if (J.format == GSLNLS_SPARSE_COO)
    {
        for (long k = 0; k < m; ++k)
        {
            ri[k] = J.i[k];
            ci[k] = J.j[k];
        }
    }
    else
        return GSLNLS_EINVAL;
    for (long k = 0; k < m; ++k)
        if (ri[k] < 0 || ri[k] >= n || ci[k] < 0 || ci[k] >= p)
            return GSLNLS_EINVAL;
    std::vector<long> ord(m);
    std::iota(ord.begin(), ord.end(), 0L);
    std::stable_sort(ord.begin(), ord.end(), [&](long a, long b) { return ri[a] != ri[b] ? ri[a] < ri[b] : ci[a] < ci[b]; });
    P.n = n;
    P.p = p;
    P.map.assign(m, 0);
    P.rowptr.assign(n + 1, 0);
    P.colidx.clear();
    std::vector<int> rows;
    long pos = -1;
    for (long q = 0; q < m; ++q)
    {
        const long e = ord[q];
        if (q == 0 || ri[e] != ri[ord[q - 1]] || ci[e] != ci[ord[q - 1]])
        {
            ++pos;
            P.colidx.push_back(ci[e]);
            rows.push_back(ri[e]);
            P.rowptr[ri[e] + 1] += 1;
        }
        P.map[e] = (int)pos;
    }
    P.nnz = pos + 1;
    for (int r = 0; r < n; ++r)
        P.rowptr[r + 1] += P.rowptr[r];
    // transpose index
    P.colptr.assign(p + 1, 0);
    for (long k = 0; k < P.nnz; ++k)
        P.colptr[P.colidx[k] + 1] += 1;
    for (int c = 0; c < p; ++c)
        P.colptr[c + 1] += P.colptr[c];
    P.rowidx.assign(P.nnz, 0);
    P.perm.assign(P.nnz, 0);
    std::vector<int> fill(P.colptr.begin(), P.colptr.end() - 1);
    for (long k = 0; k < P.nnz; ++k) // CSR order = rows ascending: CSC entries come out sorted by row
    {
        const int c = P.colidx[k], at = fill[c]++;
        P.rowidx[at] = rows[k];
        P.perm[at] = (int)k;
    }
    P.long_rows.clear();
    P.long_cols.clear();
    for (int r = 0; r < n; ++r)
        if (P.rowptr[r + 1] - P.rowptr[r] > SP_LONG)
            P.long_rows.push_back(r);
    for (int c = 0; c < p; ++c)
        if (P.colptr[c + 1] - P.colptr[c] > SP_LONG)
            P.long_cols.push_back(c);
    P.in_format = J.format;
    P.in_nnz = m;
    const int na = J.format == GSLNLS_SPARSE_COO ? (int)m : (J.format == GSLNLS_SPARSE_CSR ? n + 1 : p + 1);
    const int *a = J.format == GSLNLS_SPARSE_COO ? J.i : J.p;
    const int *b = J.format == GSLNLS_SPARSE_CSC ? J.i : J.j;
    P.in_a.assign(a, a + na);
    P.in_b.assign(b, b + m);
    return 0;
}

inline bool sparse_same_pattern(const gslnls_sparse &J, const SparsePattern &P)
{
    if (J.format != P.in_format || J.nnz != P.in_nnz)
        return false;
    const int *a = J.format == GSLNLS_SPARSE_COO ? J.i : J.p;
    const int *b = J.format == GSLNLS_SPARSE_CSC ? J.i : J.j;
    return memcmp(a, P.in_a.data(), sizeof(int) * P.in_a.size()) == 0 &&
           memcmp(b, P.in_b.data(), sizeof(int) * P.in_b.size()) == 0;
}

struct SparseCbOps : LargeOps
{
    gslnls_large_f_cb cb_f;
    gslnls_large_jac_cb cb_jac;
    void *user;
    SparsePattern pat;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // device
    double *d_y = nullptr, *d_sw = nullptr, *d_f[2] = {nullptr, nullptr}, *d_val[2] = {nullptr, nullptr},
           *d_valc[2] = {nullptr, nullptr}; // values in CSR order and, permuted once per evaluation, in CSC order
    int *d_rowptr = nullptr, *d_colidx = nullptr, *d_colptr = nullptr, *d_rowidx = nullptr, *d_perm = nullptr,
        *d_lrows = nullptr, *d_lcols = nullptr;
    double *d_vecp = nullptr, *d_outp = nullptr, *d_sqp = nullptr, *d_w = nullptr, *d_part = nullptr, *d_jtj = nullptr;
    bool jtj_current = false; // d_jtj holds J^T J of the point the driver is at
    bool dense = false;       // the callback hands over dense blocks (GSLNLS_SPARSE_DENSE): d_val[b] = n x p column-major, no index arrays
    double *d_cpart = nullptr; // scratch of the dense J^T J (partial 64 x 64 blocks of the row slices)
    size_t cpart_bytes = 0;
    long cap_nnz = 0, cap_hval = 0, dense_cap = 0;
    int cur = 0; // index of the accepted point's buffers; 1 - cur receives the trial
    // host staging in pinned memory: uploads of f (n) and the Jacobian values (nnz), p-sized vectors both ways
    double *h_f = nullptr, *h_val = nullptr, *h_pin_in = nullptr, *h_pin_out = nullptr, *h_pin_sq = nullptr;
    std::vector<double> h_part;
    static constexpr int NPART = 256;
    static_assert(NPART == SPCG_NPART, "spcg_update_kernel adds the partials of sp_sumsq_kernel");
    // device-resident CG (sparse_cg.hpp): g, diag, z, r, d, dx in one allocation of 6p doubles; scalars; pinned mirrors
    double *d_cg = nullptr, *h_pin_in2 = nullptr;
    SpCgScal *d_scal = nullptr, *h_scal = nullptr;
    int cg_last_its = 1;
    bool cg_pred_valid = false;
    double cg_pred = 0.0;

    // Allocation (streams, ~20 device and pinned buffers: ~25 ms) is separated from binding a problem: a destroyed
    // problem is parked by capi.hip and re-bound by the next gslnls_large_create_sparse of at most its size, like the
    // dense problems are (one-shot gsl_nls_large() calls on small problems were mostly allocation).
    int cap_n = 0, cap_p = 0;
    bool sw_allocated = false;
    int allocate(int n_, int p_)
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
            return GSLNLS_E_NODEVICE;
        cap_n = n_;
        cap_p = p_;
        GSLNLS_HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        GSLNLS_HIP_OK(hipEventCreate(&e0));
        GSLNLS_HIP_OK(hipEventCreate(&e1));
        const size_t nb = sizeof(double) * (size_t)cap_n, pb = sizeof(double) * (size_t)cap_p;
        GSLNLS_HIP_OK(hipMalloc(&d_y, nb));
        for (int k = 0; k < 2; ++k)
            GSLNLS_HIP_OK(hipMalloc(&d_f[k], nb));
        GSLNLS_HIP_OK(hipMalloc(&d_w, nb));
        GSLNLS_HIP_OK(hipMalloc(&d_vecp, pb));
        GSLNLS_HIP_OK(hipMalloc(&d_outp, pb));
        GSLNLS_HIP_OK(hipMalloc(&d_sqp, pb));
        GSLNLS_HIP_OK(hipMalloc(&d_part, sizeof(double) * NPART));
        GSLNLS_HIP_OK(hipMalloc(&d_rowptr, sizeof(int) * ((size_t)cap_n + 1)));
        GSLNLS_HIP_OK(hipMalloc(&d_colptr, sizeof(int) * ((size_t)cap_p + 1)));
        GSLNLS_HIP_OK(hipHostMalloc(&h_f, nb));
        GSLNLS_HIP_OK(hipHostMalloc(&h_pin_in, pb));
        GSLNLS_HIP_OK(hipHostMalloc(&h_pin_out, pb));
        GSLNLS_HIP_OK(hipHostMalloc(&h_pin_sq, pb));
        GSLNLS_HIP_OK(hipHostMalloc(&h_pin_in2, pb));
        GSLNLS_HIP_OK(hipMalloc(&d_cg, 6 * pb));
        GSLNLS_HIP_OK(hipMalloc(&d_scal, sizeof(SpCgScal)));
        GSLNLS_HIP_OK(hipHostMalloc(&h_scal, sizeof(SpCgScal)));
        h_part.resize(NPART);
        return 0;
    }
    bool fits(int n_, int p_) const { return st != nullptr && n_ <= cap_n && p_ <= cap_p; }
    // a (new) problem on the allocated buffers: data up, pattern and counters reset
    int bind(int n_, int p_, const double *y, const double *sw, gslnls_large_f_cb f, gslnls_large_jac_cb j, void *u)
    {
        n = n_;
        p = p_;
        cb_f = f;
        cb_jac = j;
        user = u;
        lazy_jac = true;
        nevalf = nevaldfu = nevaldf2 = 0;
        npass = 0;
        pass_ms = 0.f;
        cur = 0;
        cg_last_its = 1;
        cg_pred_valid = false;
        {
            // (keeps what upload_pattern allocated: cap_nnz, the value and index buffers)
            SparsePattern fresh;
            pat = fresh;
        }
        if (d_jtj)
        {
            (void)hipFree(d_jtj); // p x p of the previous problem
            d_jtj = nullptr;
            jtj_current = false;
        }
        const size_t nb = sizeof(double) * (size_t)n;
        GSLNLS_HIP_OK(hipMemcpy(d_y, y, nb, hipMemcpyHostToDevice));
        if (sw)
        {
            if (!sw_allocated)
            {
                GSLNLS_HIP_OK(hipMalloc(&d_sw_buf, sizeof(double) * (size_t)cap_n));
                sw_allocated = true;
            }
            d_sw = d_sw_buf;
            GSLNLS_HIP_OK(hipMemcpy(d_sw, sw, nb, hipMemcpyHostToDevice));
        }
        else
            d_sw = nullptr;
        const char *e = getenv("GSLNLS_LARGE_CG");
        device_cg = !(e && strcmp(e, "host") == 0);
        return 0;
    }
    double *d_sw_buf = nullptr;
    int init(int n_, int p_, const double *y, const double *sw, gslnls_large_f_cb f, gslnls_large_jac_cb j, void *u)
    {
        const int rc = allocate(n_, p_);
        return rc ? rc : bind(n_, p_, y, sw, f, j, u);
    }
    ~SparseCbOps() override
    {
        for (void *q : {(void *)d_y, (void *)d_sw_buf, (void *)d_f[0], (void *)d_f[1], (void *)d_val[0], (void *)d_val[1],
                        (void *)d_valc[0], (void *)d_valc[1], (void *)d_rowptr, (void *)d_colidx, (void *)d_colptr, (void *)d_rowidx, (void *)d_perm,
                        (void *)d_lrows, (void *)d_lcols, (void *)d_vecp, (void *)d_outp, (void *)d_sqp, (void *)d_w,
                        (void *)d_part, (void *)d_jtj, (void *)d_cg, (void *)d_scal, (void *)d_cpart})
            if (q)
                (void)hipFree(q);
        for (void *q : {(void *)h_f, (void *)h_val, (void *)h_pin_in, (void *)h_pin_out, (void *)h_pin_sq, (void *)h_pin_in2,
                        (void *)h_scal})
            if (q)
                (void)hipHostFree(q);
        if (st)
            (void)hipStreamDestroy(st);
        if (e0)
            (void)hipEventDestroy(e0);
        if (e1)
            (void)hipEventDestroy(e1);
    }
    int upload_pattern()
    {
        const long m = pat.nnz;
        if (m > cap_nnz)
        {
            for (void *q : {(void *)d_val[0], (void *)d_val[1], (void *)d_valc[0], (void *)d_valc[1], (void *)d_colidx,
                            (void *)d_rowidx, (void *)d_perm})
                if (q)
                    (void)hipFree(q);
            GSLNLS_HIP_OK(hipMalloc(&d_val[0], sizeof(double) * (size_t)m));
            GSLNLS_HIP_OK(hipMalloc(&d_val[1], sizeof(double) * (size_t)m));
            GSLNLS_HIP_OK(hipMalloc(&d_valc[0], sizeof(double) * (size_t)m));
            GSLNLS_HIP_OK(hipMalloc(&d_valc[1], sizeof(double) * (size_t)m));
            GSLNLS_HIP_OK(hipMalloc(&d_colidx, sizeof(int) * (size_t)m));
            GSLNLS_HIP_OK(hipMalloc(&d_rowidx, sizeof(int) * (size_t)m));
            GSLNLS_HIP_OK(hipMalloc(&d_perm, sizeof(int) * (size_t)m));
            cap_nnz = m;
        }
        GSLNLS_HIP_OK(hipMemcpy(d_rowptr, pat.rowptr.data(), sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice));
        GSLNLS_HIP_OK(hipMemcpy(d_colptr, pat.colptr.data(), sizeof(int) * ((size_t)p + 1), hipMemcpyHostToDevice));
        GSLNLS_HIP_OK(hipMemcpy(d_colidx, pat.colidx.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
        GSLNLS_HIP_OK(hipMemcpy(d_rowidx, pat.rowidx.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
        GSLNLS_HIP_OK(hipMemcpy(d_perm, pat.perm.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice));
        if (d_lrows)
            (void)hipFree(d_lrows);
        if (d_lcols)
            (void)hipFree(d_lcols);
        d_lrows = d_lcols = nullptr;
        if (!pat.long_rows.empty())
        {
            GSLNLS_HIP_OK(hipMalloc(&d_lrows, sizeof(int) * pat.long_rows.size()));
            GSLNLS_HIP_OK(hipMemcpy(d_lrows, pat.long_rows.data(), sizeof(int) * pat.long_rows.size(), hipMemcpyHostToDevice));
        }
        if (!pat.long_cols.empty())
        {
            GSLNLS_HIP_OK(hipMalloc(&d_lcols, sizeof(int) * pat.long_cols.size()));
            GSLNLS_HIP_OK(hipMemcpy(d_lcols, pat.long_cols.data(), sizeof(int) * pat.long_cols.size(), hipMemcpyHostToDevice));
        }
        if (!h_val || m > cap_hval)
        {
            if (h_val)
                (void)hipHostFree(h_val);
            h_val = nullptr;
            cap_hval = m > 0 ? m : 1;
            GSLNLS_HIP_OK(hipHostMalloc(&h_val, sizeof(double) * (size_t)cap_hval));
        }
        return 0;
    }
    const int *seg_flag = nullptr; // run condition of the product kernels being enqueued (device CG), else null
    int seg_run_when = 0;
    template <int L>
    void launch_segments(const int *ptr, const int *idx, const double *val, const double *vec, int nseg, const int *list,
                         int nlist, int skip_long, double *out, double *sq)
    {
        const long groups = list ? nlist : nseg;
        const long threads = groups * L;
        hipLaunchKernelGGL(sp_segment_kernel<L>, dim3((unsigned)((threads + SP_T - 1) / SP_T)), dim3(SP_T), 0, st, ptr, idx,
                           val, vec, nseg, list, nlist, skip_long, out, sq, seg_flag, seg_run_when);
    }
    // out[s] (and optionally sq[s]) over rows (transpose = false) or columns (transpose = true) of buffer b
    void segments(bool transpose, int b, const double *vec, double *out, double *sq)
    {
        if (dense)
        {
            if (transpose)
                hipLaunchKernelGGL(dn_cols_kernel, dim3(p), dim3(SP_T), 0, st, d_val[b], vec, n, out, sq, seg_flag, seg_run_when);
            else
                hipLaunchKernelGGL(dn_rows_kernel, dim3((n + SP_T - 1) / SP_T), dim3(SP_T), 0, st, d_val[b], vec, n, p, out, seg_flag,
                                   seg_run_when);
            return;
        }
        const int nseg = transpose ? p : n;
        const int *ptr = transpose ? d_colptr : d_rowptr, *idx = transpose ? d_rowidx : d_colidx;
        const double *val = transpose ? d_valc[b] : d_val[b];
        const std::vector<int> &lng = transpose ? pat.long_cols : pat.long_rows;
        const int *d_l = transpose ? d_lcols : d_lrows;
        const double mean = (double)pat.nnz / nseg;
        if (mean > 24.0)
            launch_segments<64>(ptr, idx, val, vec, nseg, nullptr, 0, 0, out, sq);
        else
        {
            if (mean > 6.0)
                launch_segments<16>(ptr, idx, val, vec, nseg, nullptr, 0, 1, out, sq);
            else if (mean > 1.5)
                launch_segments<4>(ptr, idx, val, vec, nseg, nullptr, 0, 1, out, sq);
            else
                launch_segments<1>(ptr, idx, val, vec, nseg, nullptr, 0, 1, out, sq);
            if (!lng.empty())
                launch_segments<64>(ptr, idx, val, vec, nseg, d_l, (int)lng.size(), 0, out, sq);
        }
    }
    void permute_values(int b)
    {
        if (dense)
            return; // (column-major is the order both products stream)
        hipLaunchKernelGGL(sp_permute_kernel, dim3((unsigned)((pat.nnz + SP_T - 1) / SP_T)), dim3(SP_T), 0, st, d_val[b],
                           d_perm, pat.nnz, d_valc[b]);
    }
    double sum_partials()
    {
        (void)hipMemcpyAsync(h_part.data(), d_part, sizeof(double) * NPART, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        double t = 0.0;
        for (int k = 0; k < NPART; ++k)
            t += h_part[k];
        return t;
    }
    std::vector<double> x_trial;
    // f at the trial point: callback, upload, weighting + ssr on device
    int eval(const double *x, double *ssr, double *, double *, double *, double *bad) override
    {
        const int t = 1 - cur;
        *bad = 0.0;
        x_trial.assign(x, x + p);
        if (cb_f(x, p, h_f, n, user))
            return GSLNLS_EINVAL;
        (void)hipEventRecord(e0, st);
        GSLNLS_HIP_OK(hipMemcpyAsync(d_f[t], h_f, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(sp_resid_kernel, dim3(NPART), dim3(SP_T), 0, st, d_f[t], d_y, d_sw, n, d_part);
        (void)hipEventRecord(e1, st);
        *ssr = sum_partials();
        (void)hipEventElapsedTime(&pass_ms, e0, e1);
        ++npass;
        return 0;
    }
    // J at the same point (called once the step is accepted): callback, values to the canonical layout,
    // upload, row weighting, g = J^T f and diag(J^T J) [, dense J^T J]
    int eval_jac(double *g, double *diag, double *jtj) override
    {
        const int t = 1 - cur;
        gslnls_sparse J;
        memset(&J, 0, sizeof J);
        if (cb_jac(x_trial.data(), p, &J, user))
            return GSLNLS_EINVAL;
        if (J.format == GSLNLS_SPARSE_DENSE)
        {
            // the n x p block as it is (src/nls_large.c:504, :625-633: dgemv / dsyrk on the dense J): one copy into the
            // pinned staging area, one upload; no pattern, no index arrays, no permutation
            if (J.nrow != n || J.ncol != p || !J.x)
                return GSLNLS_EINVAL;
            const long m = (long)n * p;
            if (!dense || pat.nnz != m)
            {
                dense = true;
                pat = SparsePattern();
                pat.nnz = m;
                if (m > dense_cap)
                {
                    for (void *q : {(void *)d_val[0], (void *)d_val[1]})
                        if (q)
                            (void)hipFree(q);
                    d_val[0] = d_val[1] = nullptr;
                    GSLNLS_HIP_OK(hipMalloc(&d_val[0], sizeof(double) * (size_t)m));
                    GSLNLS_HIP_OK(hipMalloc(&d_val[1], sizeof(double) * (size_t)m));
                    // (the sparse form's other arrays are allocated by upload_pattern when a sparse Jacobian arrives)
                    for (void *q : {(void *)d_valc[0], (void *)d_valc[1], (void *)d_colidx, (void *)d_rowidx, (void *)d_perm})
                        if (q)
                            (void)hipFree(q);
                    d_valc[0] = d_valc[1] = nullptr;
                    d_colidx = d_rowidx = d_perm = nullptr;
                    cap_nnz = 0; // (a sparse Jacobian later on re-allocates all of them)
                    dense_cap = m;
                }
                if (!h_val || m > cap_hval)
                {
                    if (h_val)
                        (void)hipHostFree(h_val);
                    h_val = nullptr;
                    cap_hval = m;
                    GSLNLS_HIP_OK(hipHostMalloc(&h_val, sizeof(double) * (size_t)cap_hval));
                }
            }
            memcpy(h_val, J.x, sizeof(double) * (size_t)m);
        }
        else if (dense || !sparse_same_pattern(J, pat))
        {
            if (dense)
            {
                dense = false;
                (void)hipFree(d_val[0]);
                (void)hipFree(d_val[1]);
                d_val[0] = d_val[1] = nullptr;
                cap_nnz = 0;
                dense_cap = 0;
            }
            // (a pattern that changes between points is rebuilt; the accepted point's values keep their old
            // layout, so a change is only meaningful before the first product with them -- not checked)
            const int rc = sparse_build_pattern(J, n, p, pat);
            if (rc)
                return rc;
            if (upload_pattern())
                return GSLNLS_E_NODEVICE;
        }
        if (!dense)
        {
            std::fill(h_val, h_val + pat.nnz, 0.0);
            for (long e = 0; e < J.nnz; ++e)
                h_val[pat.map[e]] += J.x[e];
        }
        (void)hipEventRecord(e0, st);
        GSLNLS_HIP_OK(hipMemcpyAsync(d_val[t], h_val, sizeof(double) * (size_t)pat.nnz, hipMemcpyHostToDevice, st));
        // (the weights scale f only: GSL's multilarge eval_f applies sqrt(w), the Jacobian callback's result is used as it
        // comes -- gsl_df_large never weights J, src/nls_large.c:629-646 -- exactly as on the row-model and wide paths
        // and in the oracle; rounds 1-2 scaled the rows of J here, which solved a different (properly weighted) problem
        // than the reference does)
        permute_values(t);
        segments(true, t, d_f[t], d_outp, d_sqp); // g = J^T f, diag(J^T J)
        GSLNLS_HIP_OK(hipMemcpyAsync(h_pin_out, d_outp, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipMemcpyAsync(h_pin_sq, d_sqp, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
        (void)hipEventRecord(e1, st);
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        memcpy(g, h_pin_out, sizeof(double) * (size_t)p);
        memcpy(diag, h_pin_sq, sizeof(double) * (size_t)p);
        (void)hipEventElapsedTime(&pass_ms, e0, e1);
        ++npass;
        if (jtj)
            return jtj_of(t, jtj);
        return 0;
    }
    void accept() override { cur = 1 - cur; }
    int jtjv(const double *, const double *u, double *normw2, double *out) override
    {
        memcpy(h_pin_in, u, sizeof(double) * (size_t)p);
        (void)hipEventRecord(e0, st);
        GSLNLS_HIP_OK(hipMemcpyAsync(d_vecp, h_pin_in, sizeof(double) * (size_t)p, hipMemcpyHostToDevice, st));
        segments(false, cur, d_vecp, d_w, nullptr);      // w = J u
        hipLaunchKernelGGL(sp_sumsq_kernel, dim3(NPART), dim3(SP_T), 0, st, d_w, n, d_part, (const int *)nullptr, 0);
        segments(true, cur, d_w, d_outp, nullptr);       // J^T w
        GSLNLS_HIP_OK(hipMemcpyAsync(h_pin_out, d_outp, sizeof(double) * (size_t)p, hipMemcpyDeviceToHost, st));
        (void)hipEventRecord(e1, st);
        *normw2 = sum_partials();
        memcpy(out, h_pin_out, sizeof(double) * (size_t)p);
        (void)hipEventElapsedTime(&pass_ms, e0, e1);
        ++npass;
        return 0;
    }
    // One Steihaug-Toint step with the recurrences on the device (sparse_cg.hpp).  Iterations are enqueued in growing
    // chunks (the first one sized by the previous step's count); behind every chunk goes the product J dx that the
    // predicted reduction will ask for, run only once the step is final; one 64-byte read-back per chunk.
    int cgst_device(const double *g, const double *diag, double delta, long cgmaxit, double *dx, int *status) override
    {
        cg_pred_valid = false;
        const size_t pb = sizeof(double) * (size_t)p;
        memcpy(h_pin_in, g, pb);
        memcpy(h_pin_in2, diag, pb);
        SpCgVecs v;
        double *base = d_cg;
        GSLNLS_HIP_OK(hipMemcpyAsync(base, h_pin_in, pb, hipMemcpyHostToDevice, st));
        GSLNLS_HIP_OK(hipMemcpyAsync(base + p, h_pin_in2, pb, hipMemcpyHostToDevice, st));
        v.g = base;
        v.diag = base + (size_t)p;
        v.z = base + 2 * (size_t)p;
        v.r = base + 3 * (size_t)p;
        v.d = base + 4 * (size_t)p;
        v.dx = base + 5 * (size_t)p;
        v.u = d_vecp;
        v.Bd = d_outp;
        v.part = d_part;
        v.s = d_scal;
        v.p = p;
        v.cgmaxit = cgmaxit;
        const int T = p <= 8192 ? 256 : 1024;
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(spcg_init_kernel, dim3(1), dim3(T), 0, st, v, delta);
        const bool small = p <= 65536; // the step rides along with every read-back when it is small
        int chunk = std::min(32, std::max(2, cg_last_its + 1));
        for (;;)
        {
            seg_flag = &d_scal->done;
            seg_run_when = 0;
            for (int k = 0; k < chunk; ++k)
            {
                segments(false, cur, d_vecp, d_w, nullptr); // w = J u
                hipLaunchKernelGGL(sp_sumsq_kernel, dim3(NPART), dim3(SP_T), 0, st, d_w, n, d_part, seg_flag, 0);
                segments(true, cur, d_w, d_outp, nullptr);  // J^T w
                hipLaunchKernelGGL(spcg_update_kernel, dim3(1), dim3(T), 0, st, v);
            }
            seg_run_when = 1; // the finished step's product with J (u = dx by then)
            segments(false, cur, d_vecp, d_w, nullptr);
            hipLaunchKernelGGL(sp_sumsq_kernel, dim3(NPART), dim3(SP_T), 0, st, d_w, n, d_part, seg_flag, 1);
            hipLaunchKernelGGL(spcg_pred_kernel, dim3(1), dim3(64), 0, st, v);
            seg_flag = nullptr;
            GSLNLS_HIP_OK(hipMemcpyAsync(h_scal, d_scal, sizeof(SpCgScal), hipMemcpyDeviceToHost, st));
            if (small)
                GSLNLS_HIP_OK(hipMemcpyAsync(h_pin_out, v.dx, pb, hipMemcpyDeviceToHost, st));
            (void)hipEventRecord(e1, st);
            GSLNLS_HIP_OK(hipStreamSynchronize(st));
            if (h_scal->done)
                break;
            chunk = std::min(32, chunk * 2);
        }
        if (!small)
        {
            GSLNLS_HIP_OK(hipMemcpyAsync(h_pin_out, v.dx, pb, hipMemcpyDeviceToHost, st));
            GSLNLS_HIP_OK(hipStreamSynchronize(st));
        }
        memcpy(dx, h_pin_out, pb);
        (void)hipEventElapsedTime(&pass_ms, e0, e1);
        nevaldfu += (long)(h_scal->n_notrans + h_scal->n_trans);
        npass += (int)h_scal->it;
        cg_last_its = (int)std::min<long long>(h_scal->it, 64);
        cg_pred_valid = h_scal->njdx2_valid != 0;
        cg_pred = h_scal->njdx2;
        *status = h_scal->status;
        return 0;
    }
    bool cached_njdx2(double *v) override
    {
        if (!cg_pred_valid)
            return false;
        *v = cg_pred;
        cg_pred_valid = false;
        ++npass;
        return true;
    }
    int jtj_of(int b, double *jtj)
    {
        const size_t bytes = sizeof(double) * (size_t)p * p;
        if (!d_jtj)
            GSLNLS_HIP_OK(hipMalloc(&d_jtj, bytes));
        if (dense)
        {
            if (const int rc = bd_dense_jtj(d_val[b], n, p, d_jtj, st, &d_cpart, &cpart_bytes))
                return rc;
        }
        else
        {
            GSLNLS_HIP_OK(hipMemsetAsync(d_jtj, 0, bytes, st));
            const long threads = 64L * p;
            hipLaunchKernelGGL(sp_jtj_kernel, dim3((unsigned)((threads + SP_T - 1) / SP_T)), dim3(SP_T), 0, st, d_colptr, d_rowidx,
                               d_perm, d_rowptr, d_colidx, d_val[b], p, d_jtj);
        }
        jtj_current = false;
        if (!jtj_device_only)
            GSLNLS_HIP_OK(hipMemcpyAsync(jtj, d_jtj, bytes, hipMemcpyDeviceToHost, st));
        GSLNLS_HIP_OK(hipStreamSynchronize(st));
        jtj_current = true;
        ++npass;
        return 0;
    }
    bool can_keep_jtj_on_device() const override { return true; }
    int jtj_download(double *jtj) override
    {
        if (!jtj_current || !d_jtj)
            return GSLNLS_FAILURE;
        GSLNLS_HIP_OK(hipMemcpy(jtj, d_jtj, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost));
        return GSLNLS_SUCCESS;
    }
    int full_jtj(const double *, double *jtj) override { return jtj_of(cur, jtj); }
    // (eval_jac runs at accepted points only and ends in a stream synchronisation: what d_jtj holds is the current point's)
    const double *jtj_device() override { return jtj_current ? d_jtj : nullptr; }
    int residual(const double *, double *resid_host) override
    {
        GSLNLS_HIP_OK(hipMemcpy(resid_host, d_f[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
        return 0;
    }
};

} // namespace gslnls
