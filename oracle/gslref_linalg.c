/*
 * gslref_linalg.c -- ORACLE (test infrastructure, never shipped): dense linear
 * algebra that the reference reaches through GSL (un-vendored, GSL >= 2.3):
 *   gsl_linalg_mcholesky_decomp/_solve  (solver="cholesky": called through
 *       params->solver->presolve/solve, src/trust.c:237-248)
 *   gsl_linalg_cholesky_decomp1/_invert (src/nls_utils.c:64, :92)
 *   QR least squares of [J; sqrt(mu) D]  (solver="qr", R default R/nls.R:1186)
 *   gsl_multifit_nlinear_covar           (src/nls.c:607)
 * All matrices here are row-major.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "gslref_internal.h"

/* ------------------------------------------------------------------ */
/* modified Cholesky, Gill-Murray-Wright with diagonal pivoting:
 * P (A + E) P^T = L D L^T.   GSL: linalg/mcholesky.c                  */
static void sym_swap_rowcol(int N, double *A, int i, int j)
{
    /* symmetric interchange of rows/cols i<j of a matrix held in the lower triangle */
    int k;
    double t;
    if (i == j)
        return;
    if (i > j)
    {
        int s = i;
        i = j;
        j = s;
    }
    /* diagonal */
    t = A[i * N + i];
    A[i * N + i] = A[j * N + j];
    A[j * N + j] = t;
    /* left parts of rows i and j: columns k < i */
    for (k = 0; k < i; ++k)
    {
        t = A[i * N + k];
        A[i * N + k] = A[j * N + k];
        A[j * N + k] = t;
    }
    /* column i below i up to j, vs row j between i and j */
    for (k = i + 1; k < j; ++k)
    {
        t = A[k * N + i];
        A[k * N + i] = A[j * N + k];
        A[j * N + k] = t;
    }
    /* below j: columns i and j */
    for (k = j + 1; k < N; ++k)
    {
        t = A[k * N + i];
        A[k * N + i] = A[k * N + j];
        A[k * N + j] = t;
    }
    /* element (j,i) stays */
}

int gslref_mcholesky_decomp(int N, double *A, int *perm)
{
    const double delta = DBL_EPSILON;
    double beta, gamma = 0.0, xi = 0.0;
    int i, j, k;

    for (i = 0; i < N; ++i)
        perm[i] = i;

    for (i = 0; i < N; ++i)
    {
        gamma = fmax(gamma, fabs(A[i * N + i]));
        for (j = 0; j < i; ++j)
            xi = fmax(xi, fabs(A[i * N + j]));
    }
    if (N == 1)
        beta = fmax(fmax(gamma, xi), DBL_EPSILON);
    else
    {
        double nu = sqrt((double)N * N - 1.0);
        beta = fmax(fmax(gamma, xi / nu), DBL_EPSILON);
    }
    beta = sqrt(beta);

    for (j = 0; j < N; ++j)
    {
        double ajj, thetaj = 0.0, u, alpha, alphainv;
        int q = j;
        double maxd = fabs(A[j * N + j]);
        for (i = j + 1; i < N; ++i)
        {
            double d = fabs(A[i * N + i]);
            if (d > maxd)
            {
                maxd = d;
                q = i;
            }
        }
        if (q != j)
        {
            int t = perm[q];
            perm[q] = perm[j];
            perm[j] = t;
            sym_swap_rowcol(N, A, q, j);
        }
        for (i = j + 1; i < N; ++i)
            thetaj = fmax(thetaj, fabs(A[i * N + j]));
        u = thetaj / beta;
        ajj = A[j * N + j];
        alpha = fmax(fmax(delta, fabs(ajj)), u * u);
        alphainv = 1.0 / alpha;
        /* trailing update m -= v v^T / alpha (lower), then v /= alpha */
        for (i = j + 1; i < N; ++i)
        {
            double vi = A[i * N + j];
            for (k = j + 1; k <= i; ++k)
                A[i * N + k] -= alphainv * vi * A[k * N + j];
        }
        for (i = j + 1; i < N; ++i)
            A[i * N + j] *= alphainv;
        A[j * N + j] = alpha;
    }
    return GSLREF_SUCCESS;
}

int gslref_mcholesky_solve(int N, const double *LDLT, const int *perm, const double *b, double *x)
{
    int i, j;
    double *t = (double *)malloc(sizeof(double) * N);
    for (i = 0; i < N; ++i)
        t[i] = b[perm[i]];
    /* L z = Pb */
    for (i = 0; i < N; ++i)
        for (j = 0; j < i; ++j)
            t[i] -= LDLT[i * N + j] * t[j];
    for (i = 0; i < N; ++i)
        t[i] /= LDLT[i * N + i];
    /* L^T w = z */
    for (i = N - 1; i >= 0; --i)
        for (j = i + 1; j < N; ++j)
            t[i] -= LDLT[j * N + i] * t[j];
    for (i = 0; i < N; ++i)
        x[perm[i]] = t[i];
    free(t);
    return GSLREF_SUCCESS;
}

/* ------------------------------------------------------------------ */
/* plain Cholesky A = L L^T in the lower triangle (gsl_linalg_cholesky_decomp1);
 * returns GSLREF_FAILURE (GSL: GSL_EDOM) when not positive definite. */
int gslref_cholesky_decomp1(int N, double *A)
{
    int i, j, k;
    for (j = 0; j < N; ++j)
    {
        double ajj = A[j * N + j];
        for (k = 0; k < j; ++k)
            ajj -= A[j * N + k] * A[j * N + k];
        if (!(ajj > 0.0))
            return GSLREF_FAILURE;
        ajj = sqrt(ajj);
        A[j * N + j] = ajj;
        for (i = j + 1; i < N; ++i)
        {
            double s = A[i * N + j];
            for (k = 0; k < j; ++k)
                s -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = s / ajj;
        }
    }
    return GSLREF_SUCCESS;
}

/* (L L^T)^{-1} from the Cholesky factor, full symmetric result (gsl_linalg_cholesky_invert) */
int gslref_cholesky_invert(int N, double *A)
{
    int i, j, k;
    double *Li = (double *)calloc((size_t)N * N, sizeof(double));
    /* Li = L^{-1} (lower) */
    for (j = 0; j < N; ++j)
    {
        if (A[j * N + j] == 0.0)
        {
            free(Li);
            return GSLREF_FAILURE;
        }
        Li[j * N + j] = 1.0 / A[j * N + j];
        for (i = j + 1; i < N; ++i)
        {
            double s = 0.0;
            for (k = j; k < i; ++k)
                s -= A[i * N + k] * Li[k * N + j];
            Li[i * N + j] = s / A[i * N + i];
        }
    }
    /* A^{-1} = Li^T Li */
    for (i = 0; i < N; ++i)
        for (j = 0; j <= i; ++j)
        {
            double s = 0.0;
            for (k = i; k < N; ++k)
                s += Li[k * N + i] * Li[k * N + j];
            A[i * N + j] = s;
            A[j * N + i] = s;
        }
    free(Li);
    return GSLREF_SUCCESS;
}

/* src/nls_utils.c:55-73 */
double gslref_det_cholesky_jtj(int n, int p, const double *J)
{
    double det = 0.0;
    int i, j, k;
    double *JTJ = (double *)calloc((size_t)p * p, sizeof(double));
    for (k = 0; k < n; ++k)
        for (i = 0; i < p; ++i)
        {
            double jki = J[(size_t)k * p + i];
            for (j = 0; j <= i; ++j)
                JTJ[i * p + j] += jki * J[(size_t)k * p + j];
        }
    if (gslref_cholesky_decomp1(p, JTJ) == GSLREF_SUCCESS)
    {
        det = 1.0;
        for (i = 0; i < p; ++i)
            det *= JTJ[i * p + i];
        det = det * det;
    }
    free(JTJ);
    return det;
}

/* ------------------------------------------------------------------ */
/* Householder QR with column pivoting of an m x p row-major matrix A (m >= p).
 * On exit R is in the upper triangle, Householder vectors below, tau[p],
 * perm[p] (column k of AP is column perm[k] of A). */
static void qrpt_decomp(int m, int p, double *A, double *tau, int *perm)
{
    int i, j, k;
    double *cn = (double *)malloc(sizeof(double) * p);
    for (j = 0; j < p; ++j)
    {
        double s = 0.0;
        for (i = 0; i < m; ++i)
            s += A[(size_t)i * p + j] * A[(size_t)i * p + j];
        cn[j] = s;
        perm[j] = j;
    }
    for (k = 0; k < p; ++k)
    {
        /* pivot: column of largest remaining norm (recomputed for robustness) */
        int piv = k;
        double best = -1.0;
        for (j = k; j < p; ++j)
        {
            double s = 0.0;
            for (i = k; i < m; ++i)
                s += A[(size_t)i * p + j] * A[(size_t)i * p + j];
            cn[j] = s;
            if (s > best)
            {
                best = s;
                piv = j;
            }
        }
        if (piv != k)
        {
            int t = perm[piv];
            perm[piv] = perm[k];
            perm[k] = t;
            for (i = 0; i < m; ++i)
            {
                double v = A[(size_t)i * p + piv];
                A[(size_t)i * p + piv] = A[(size_t)i * p + k];
                A[(size_t)i * p + k] = v;
            }
        }
        {
            double alpha = A[(size_t)k * p + k];
            double xnorm = 0.0, beta, t;
            for (i = k + 1; i < m; ++i)
                xnorm += A[(size_t)i * p + k] * A[(size_t)i * p + k];
            xnorm = sqrt(xnorm);
            if (xnorm == 0.0)
            {
                tau[k] = 0.0;
                continue;
            }
            beta = -(alpha >= 0.0 ? 1.0 : -1.0) * hypot(alpha, xnorm);
            t = (beta - alpha) / beta;
            {
                double s = 1.0 / (alpha - beta);
                for (i = k + 1; i < m; ++i)
                    A[(size_t)i * p + k] *= s;
            }
            A[(size_t)k * p + k] = beta;
            tau[k] = t;
            /* apply H = I - tau v v^T to remaining columns */
            for (j = k + 1; j < p; ++j)
            {
                double w = A[(size_t)k * p + j];
                for (i = k + 1; i < m; ++i)
                    w += A[(size_t)i * p + k] * A[(size_t)i * p + j];
                w *= t;
                A[(size_t)k * p + j] -= w;
                for (i = k + 1; i < m; ++i)
                    A[(size_t)i * p + j] -= w * A[(size_t)i * p + k];
            }
        }
    }
    free(cn);
}

static void qr_qtvec(int m, int p, const double *QR, const double *tau, double *b)
{
    int i, k;
    for (k = 0; k < p; ++k)
    {
        double w;
        if (tau[k] == 0.0)
            continue;
        w = b[k];
        for (i = k + 1; i < m; ++i)
            w += QR[(size_t)i * p + k] * b[i];
        w *= tau[k];
        b[k] -= w;
        for (i = k + 1; i < m; ++i)
            b[i] -= w * QR[(size_t)i * p + k];
    }
}

/* min || A x - b ||, A is m x p row-major (destroyed), b length m (destroyed).
 * Rank-revealing: tiny pivots give zero components (minimum-norm flavour of
 * gsl_linalg_QRPT_lssolve2 as used by GSL's qr.c when mu == 0). */
int gslref_lstsq(int m, int p, double *A, double *b, double *x)
{
    double *tau = (double *)malloc(sizeof(double) * p);
    int *perm = (int *)malloc(sizeof(int) * p);
    double *z = (double *)calloc(p, sizeof(double));
    int i, j, rank = p;
    double r00;
    qrpt_decomp(m, p, A, tau, perm);
    qr_qtvec(m, p, A, tau, b);
    r00 = fabs(A[0]);
    for (i = 0; i < p; ++i)
        if (!(fabs(A[(size_t)i * p + i]) > DBL_EPSILON * p * r00))
        {
            rank = i;
            break;
        }
    for (i = rank - 1; i >= 0; --i)
    {
        double s = b[i];
        for (j = i + 1; j < rank; ++j)
            s -= A[(size_t)i * p + j] * z[j];
        z[i] = s / A[(size_t)i * p + i];
    }
    for (i = 0; i < p; ++i)
        x[perm[i]] = z[i];
    free(tau);
    free(perm);
    free(z);
    return GSLREF_SUCCESS;
}

/* gsl_multifit_nlinear_covar(J, 0.0, covar): (J^T J)^{-1} through a pivoted QR of J
 * (GSL multifit_nlinear/covar.c, MINPACK covar).  Rank-deficient trailing block
 * gets zeros like MINPACK.  cov is p x p row-major (symmetric). */
int gslref_covar(int n, int p, const double *J, double *cov)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)n * p);
    double *tau = (double *)malloc(sizeof(double) * p);
    int *perm = (int *)malloc(sizeof(int) * p);
    double *Ri = (double *)calloc((size_t)p * p, sizeof(double));
    double *C = (double *)calloc((size_t)p * p, sizeof(double));
    int i, j, k, kmax = p;
    memcpy(A, J, sizeof(double) * (size_t)n * p);
    qrpt_decomp(n, p, A, tau, perm);
    /* epsrel = 0: tolr = 0, kmax = first k with R_kk == 0 */
    for (k = 0; k < p && k < n; ++k)
        if (A[(size_t)k * p + k] == 0.0)
        {
            kmax = k;
            break;
        }
    if (n < p && kmax > n)
        kmax = n;
    /* Ri = R^{-1} (upper, leading kmax block) */
    for (j = 0; j < kmax; ++j)
    {
        Ri[j * p + j] = 1.0 / A[(size_t)j * p + j];
        for (i = j - 1; i >= 0; --i)
        {
            double s = 0.0;
            for (k = i + 1; k <= j; ++k)
                s -= A[(size_t)i * p + k] * Ri[k * p + j];
            Ri[i * p + j] = s / A[(size_t)i * p + i];
        }
    }
    /* C = Ri Ri^T in permuted coordinates */
    for (i = 0; i < kmax; ++i)
        for (j = 0; j < kmax; ++j)
        {
            double s = 0.0;
            int k0 = i > j ? i : j;
            for (k = k0; k < kmax; ++k)
                s += Ri[i * p + k] * Ri[j * p + k];
            C[i * p + j] = s;
        }
    for (i = 0; i < p; ++i)
        for (j = 0; j < p; ++j)
            cov[perm[i] * p + perm[j]] = C[i * p + j];
    free(A);
    free(tau);
    free(perm);
    free(Ri);
    free(C);
    return GSLREF_SUCCESS;
}
