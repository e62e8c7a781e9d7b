/*
 * Plain-C float64 ORACLE for the tet-sphere geometry energy -- TEST INFRASTRUCTURE.
 *
 * Second, independent restatement of the reference algorithm (the first is
 * oracle/tet_energy_oracle.py); the two are cross-checked in tests/ and this
 * one is fast enough to check million-tet GPU results.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (tssplat_amd/csrc) shares no code with this file.
 *
 * Follows (citations into /root/reference):
 *   tssplat_ext/tet_spheres/tet_spheres_cuda.cu:21-30   det (six-term expansion)
 *   tssplat_ext/tet_spheres/tet_spheres_cuda.cu:32-46   ddetA_dA (cofactor)
 *   tssplat_ext/tet_spheres/tet_spheres_cuda.cu:48-66   penalty forward, order 2|4 else 0
 *   tssplat_ext/tet_spheres/tet_spheres_cuda.cu:68-102  penalty backward
 *   tssplat_ext/tet_spheres/tet_spheres_cuda.cu:118-195 E = c1*0.5*x'Mx + c2*sum pen
 *   tssplat_ext/tet_spheres/tet_spheres_cuda.cu:197-263 g = gradH*(c1*Mx + c2*G' dpen)
 *   tssplat_ext/tet_spheres/tet_spheres.cpp:148-149     M = G'L'LG (libpgo), G
 *   tssplat_ext/tet_spheres/tet_spheres.cpp:252-255     fp32 -> double rest positions
 *   tssplat_ext/tet_spheres/tet_spheres.cpp:43-45       double -> fp32 operator values
 *   geometry/mesh_utils.py:38-69                        G: F = Ds * Dm^-1
 *
 * PARITY UNPINNED for L: libpgo's pgo_create_tet_biharmonic_gradient_matrix
 * (faceNeighbor=1, scale=0) is not vendored and has no golden values in the
 * reference; it is restated as the uniform face-adjacency graph Laplacian over
 * tets, (LF)_e = deg(e) F_e - sum_{e'~e} F_e'.  G is pinned by
 * tests/golden/g_matrix_golden.npz (generated from the reference's own
 * compute_G_matrix).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t a, b, c;   /* sorted face */
    int32_t slot;      /* 4*tet + local opposite vertex */
} tso_face;

static int face_cmp(const void *pa, const void *pb)
{
    const tso_face *x = (const tso_face *)pa, *y = (const tso_face *)pb;
    if (x->a != y->a) return x->a < y->a ? -1 : 1;
    if (x->b != y->b) return x->b < y->b ? -1 : 1;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    return 0;
}

static void sort3(int32_t *v)
{
    int32_t t;
    if (v[0] > v[1]) { t = v[0]; v[0] = v[1]; v[1] = t; }
    if (v[1] > v[2]) { t = v[1]; v[1] = v[2]; v[2] = t; }
    if (v[0] > v[1]) { t = v[0]; v[0] = v[1]; v[1] = t; }
}

/* nbr[4*e+k] = tet across the face of e opposite local vertex k, or -1.
 * returns 0, 1 on allocation failure, 2 on a face shared by >2 tets. */
int tso_face_adjacency(int64_t m, const int32_t *tets, int32_t *nbr)
{
    static const int opp[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    int64_t nf = 4 * m, i;
    tso_face *f = (tso_face *)malloc(sizeof(tso_face) * (size_t)(nf ? nf : 1));
    if (!f) return 1;
    for (i = 0; i < m; i++) {
        int k;
        for (k = 0; k < 4; k++) {
            int32_t v[3];
            v[0] = tets[4 * i + opp[k][0]];
            v[1] = tets[4 * i + opp[k][1]];
            v[2] = tets[4 * i + opp[k][2]];
            sort3(v);
            f[4 * i + k].a = v[0]; f[4 * i + k].b = v[1]; f[4 * i + k].c = v[2];
            f[4 * i + k].slot = (int32_t)(4 * i + k);
        }
    }
    qsort(f, (size_t)nf, sizeof(tso_face), face_cmp);
    for (i = 0; i < nf; i++) nbr[i] = -1;
    for (i = 0; i + 1 < nf; i++) {
        if (face_cmp(&f[i], &f[i + 1]) == 0) {
            if (i + 2 < nf && face_cmp(&f[i], &f[i + 2]) == 0) { free(f); return 2; }
            nbr[f[i].slot] = f[i + 1].slot / 4;
            nbr[f[i + 1].slot] = f[i].slot / 4;
            i++;
        }
    }
    free(f);
    return 0;
}

static double det3(const double F[9])
{
    /* row-major F[3*i+j]; same six terms as .cu:24-29 */
    return -F[2] * F[4] * F[6] + F[1] * F[5] * F[6] + F[2] * F[3] * F[7]
           - F[0] * F[5] * F[7] - F[1] * F[3] * F[8] + F[0] * F[4] * F[8];
}

static void cof3(const double F[9], double C[9])
{
    C[0] = F[4] * F[8] - F[5] * F[7];
    C[1] = F[5] * F[6] - F[3] * F[8];
    C[2] = F[3] * F[7] - F[4] * F[6];
    C[3] = F[2] * F[7] - F[1] * F[8];
    C[4] = F[0] * F[8] - F[2] * F[6];
    C[5] = F[1] * F[6] - F[0] * F[7];
    C[6] = F[1] * F[5] - F[2] * F[4];
    C[7] = F[2] * F[3] - F[0] * F[5];
    C[8] = F[0] * F[4] - F[1] * F[3];
}

/* Dm^-1 per tet, double from float32 rest positions; optionally rounded to fp32.
 * returns 0 or 3 if a rest tet is singular. */
int tso_rest_inverse(int64_t m, const float *rest, const int32_t *tets, int round_fp32, double *dminv)
{
    int64_t e;
    for (e = 0; e < m; e++) {
        const int32_t *t = tets + 4 * e;
        double D[9], C[9], d;
        int i, k;
        for (i = 0; i < 3; i++)
            for (k = 0; k < 3; k++)
                D[3 * i + k] = (double)rest[3 * t[k + 1] + i] - (double)rest[3 * t[0] + i];
        d = det3(D);
        if (d == 0.0 || !isfinite(d)) return 3;
        cof3(D, C);
        /* inverse = cof^T / det */
        for (i = 0; i < 3; i++)
            for (k = 0; k < 3; k++) {
                double v = C[3 * k + i] / d;
                if (round_fp32) v = (double)(float)v;
                dminv[9 * e + 3 * i + k] = v;
            }
    }
    return 0;
}

/*
 * E_out = {E, E_s, E_b}; grad (3n doubles) may be NULL.  nbr may be NULL (built here).
 * returns 0 on success.
 */
int tso_energy_grad(int64_t n, int64_t m, const float *rest, const int32_t *tets, const int32_t *nbr_in,
                    const float *x, float c1f, float c2f, int order, double grad_output,
                    double *E_out, double *grad)
{
    double c1 = (double)c1f, c2 = (double)c2f;
    double *dminv = NULL, *F = NULL, *H = NULL;
    int32_t *nbr = NULL;
    int64_t e;
    int rc = 0, a, i, j, k;
    double Es = 0.0, Eb = 0.0;

    dminv = (double *)malloc(sizeof(double) * 9 * (size_t)(m ? m : 1));
    F = (double *)malloc(sizeof(double) * 9 * (size_t)(m ? m : 1));
    H = (double *)malloc(sizeof(double) * 9 * (size_t)(m ? m : 1));
    nbr = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(m ? m : 1));
    if (!dminv || !F || !H || !nbr) { rc = 1; goto done; }
    if (nbr_in) memcpy(nbr, nbr_in, sizeof(int32_t) * 4 * (size_t)m);
    else if ((rc = tso_face_adjacency(m, tets, nbr)) != 0) goto done;
    if ((rc = tso_rest_inverse(m, rest, tets, 1, dminv)) != 0) goto done;

    for (e = 0; e < m; e++) {                 /* F = Ds * Dm^-1 */
        const int32_t *t = tets + 4 * e;
        double Ds[9];
        for (i = 0; i < 3; i++)
            for (k = 0; k < 3; k++)
                Ds[3 * i + k] = (double)x[3 * t[k + 1] + i] - (double)x[3 * t[0] + i];
        for (i = 0; i < 3; i++)
            for (j = 0; j < 3; j++) {
                double s = 0.0;
                for (k = 0; k < 3; k++) s += Ds[3 * i + k] * dminv[9 * e + 3 * k + j];
                F[9 * e + 3 * i + j] = s;
            }
    }
    for (e = 0; e < m; e++) {                 /* H = L F, E_s, E_b */
        int deg = 0;
        double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, J, Jm;
        for (k = 0; k < 4; k++) {
            int32_t q = nbr[4 * e + k];
            if (q < 0) continue;
            deg++;
            for (i = 0; i < 9; i++) h[i] -= F[9 * q + i];
        }
        for (i = 0; i < 9; i++) {
            h[i] += deg * F[9 * e + i];
            H[9 * e + i] = h[i];
            Es += 0.5 * h[i] * h[i];
        }
        J = det3(F + 9 * e);
        Jm = J < 0 ? -J : 0.0;
        if (order == 2) Eb += Jm * Jm;
        else if (order == 4) Eb += Jm * Jm * Jm * Jm;
    }
    E_out[0] = c1 * Es + c2 * Eb;
    E_out[1] = Es;
    E_out[2] = Eb;

    if (grad) {
        memset(grad, 0, sizeof(double) * 3 * (size_t)n);
        for (e = 0; e < m; e++) {
            const int32_t *t = tets + 4 * e;
            int deg = 0;
            double P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, J, d[9];
            for (k = 0; k < 4; k++) {         /* Q = L^T H (L symmetric) */
                int32_t q = nbr[4 * e + k];
                if (q < 0) continue;
                deg++;
                for (i = 0; i < 9; i++) P[i] -= H[9 * q + i];
            }
            for (i = 0; i < 9; i++) P[i] = c1 * (P[i] + deg * H[9 * e + i]);
            J = det3(F + 9 * e);
            if (J < 0) {
                double C[9], Jm = -J, s = 0.0;
                cof3(F + 9 * e, C);
                if (order == 2) s = 2.0 * Jm;
                else if (order == 4) s = 4.0 * Jm * Jm * Jm;
                for (i = 0; i < 9; i++) P[i] += c2 * (-s) * C[i];
            }
            /* dE/dDs = P * Dm^-T : d[i][k] = sum_j P[i][j] Dminv[k][j] */
            for (i = 0; i < 3; i++)
                for (k = 0; k < 3; k++) {
                    double s = 0.0;
                    for (j = 0; j < 3; j++) s += P[3 * i + j] * dminv[9 * e + 3 * k + j];
                    d[3 * i + k] = s;
                }
            for (i = 0; i < 3; i++) {
                double tot = 0.0;
                for (k = 0; k < 3; k++) {
                    grad[3 * t[k + 1] + i] += d[3 * i + k];
                    tot += d[3 * i + k];
                }
                grad[3 * t[0] + i] -= tot;
            }
        }
        if (grad_output != 1.0)
            for (e = 0; e < 3 * n; e++) grad[e] *= grad_output;
    }
    (void)a;
done:
    free(dminv); free(F); free(H); free(nbr);
    return rc;
}

/*
 * Statistical model of the fp32 rounding error of a FACTORED evaluation -- plain-C twin of
 * oracle/tet_energy_oracle.py::rounding_error_model (same formulas, uniform element Laplacian), fast enough for
 * the 21 M-tet scene.  Every fp32 operation gets an independent relative error of standard deviation u = 2^-24;
 * variances are pushed through F = Ds Dm^-1, H = L F, Q = L^T H, P = c1 Q + c2 pen' cof F, d = P Dm^-T and the
 * per-vertex sums.  out_std_E: predicted standard deviation (+ second-order bias) of the energy error;
 * out_std_g[n]: predicted standard deviation of every vertex' gradient error (Euclidean norm, grad_output = 1).
 * Not a restatement of reference code: it is the yardstick the GPU parity tests hold the measured errors against.
 */
int tso_rounding_model(int64_t n, int64_t m, const float *rest, const int32_t *tets, const int32_t *nbr_in,
                       const float *x, float c1f, float c2f, int order, double *out_std_E, double *out_std_g)
{
    const double u2 = ldexp(1.0, -48);
    const double c1 = (double)c1f, c2 = (double)c2f;
    double *dminv = NULL, *F = NULL, *H = NULL;
    float *vF = NULL, *vH = NULL;
    int32_t *nbr = NULL;
    int64_t e;
    int rc = 0, i, j, k;
    double Es = 0.0, Eb = 0.0, sHHvH = 0.0, sH4 = 0.0, sdpvJ = 0.0, spen2 = 0.0, svH = 0.0;

    dminv = (double *)malloc(sizeof(double) * 9 * (size_t)(m ? m : 1));
    F = (double *)malloc(sizeof(double) * 9 * (size_t)(m ? m : 1));
    H = (double *)malloc(sizeof(double) * 9 * (size_t)(m ? m : 1));
    vF = (float *)malloc(sizeof(float) * 9 * (size_t)(m ? m : 1));
    vH = (float *)malloc(sizeof(float) * 9 * (size_t)(m ? m : 1));
    nbr = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(m ? m : 1));
    if (!dminv || !F || !H || !vF || !vH || !nbr) { rc = 1; goto done; }
    if (nbr_in) memcpy(nbr, nbr_in, sizeof(int32_t) * 4 * (size_t)m);
    else if ((rc = tso_face_adjacency(m, tets, nbr)) != 0) goto done;
    if ((rc = tso_rest_inverse(m, rest, tets, 1, dminv)) != 0) goto done;

#pragma omp parallel for private(i, j, k) schedule(static)
    for (e = 0; e < m; e++) {                 /* F and var F */
        const int32_t *t = tets + 4 * e;
        double Ds[9];
        for (i = 0; i < 3; i++)
            for (k = 0; k < 3; k++)
                Ds[3 * i + k] = (double)x[3 * t[k + 1] + i] - (double)x[3 * t[0] + i];
        for (i = 0; i < 3; i++)
            for (j = 0; j < 3; j++) {
                double s = 0.0, v = 0.0;
                for (k = 0; k < 3; k++) {
                    const double p = Ds[3 * i + k] * dminv[9 * e + 3 * k + j];
                    s += p;
                    v += p * p;
                }
                F[9 * e + 3 * i + j] = s;
                vF[9 * e + 3 * i + j] = (float)(u2 * v);
            }
    }
#pragma omp parallel for private(i, k) reduction(+ : Es, sHHvH, sH4, svH) schedule(static)
    for (e = 0; e < m; e++) {                 /* H = L F, var H (L o L: deg^2 on the diagonal, 1 off it) */
        int deg = 0;
        double h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (k = 0; k < 4; k++) {
            int32_t q = nbr[4 * e + k];
            if (q < 0) continue;
            deg++;
            for (i = 0; i < 9; i++) {
                h[i] -= F[9 * q + i];
                v[i] += (double)vF[9 * q + i] + u2 * F[9 * q + i] * F[9 * q + i];
            }
        }
        for (i = 0; i < 9; i++) {
            const double f = F[9 * e + i], d2 = (double)deg * deg;
            h[i] += deg * f;
            v[i] += d2 * ((double)vF[9 * e + i] + u2 * f * f);
            H[9 * e + i] = h[i];
            vH[9 * e + i] = (float)v[i];
            Es += 0.5 * h[i] * h[i];
            sHHvH += h[i] * h[i] * v[i];
            sH4 += h[i] * h[i] * h[i] * h[i];
            svH += v[i];
        }
    }
    for (i = 0; i < n; i++) out_std_g[i] = 0.0;
#pragma omp parallel for private(i, j, k) reduction(+ : Eb, sdpvJ, spen2) schedule(static)
    for (e = 0; e < m; e++) {
        const int32_t *t = tets + 4 * e;
        const double *f = F + 9 * e, *di = dminv + 9 * e;
        int deg = 0;
        double Q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, vQ[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double C[9], vC[9], J, Jm, pen = 0.0, dpen = 0.0, ddpen = 0.0, vJ = 0.0, Pm[9], vP[9], d[9], vd[9];
        for (k = 0; k < 4; k++) {
            int32_t q = nbr[4 * e + k];
            if (q < 0) continue;
            deg++;
            for (i = 0; i < 9; i++) {
                Q[i] -= H[9 * q + i];
                vQ[i] += (double)vH[9 * q + i] + u2 * H[9 * q + i] * H[9 * q + i];
            }
        }
        for (i = 0; i < 9; i++) {
            const double h = H[9 * e + i], d2 = (double)deg * deg;
            Q[i] += deg * h;
            vQ[i] += d2 * ((double)vH[9 * e + i] + u2 * h * h);
        }
        J = det3(f);
        Jm = J < 0 ? -J : 0.0;
        if (order == 2) { pen = Jm * Jm; dpen = J < 0 ? -2.0 * Jm : 0.0; ddpen = J < 0 ? 2.0 : 0.0; }
        else if (order == 4) { pen = Jm * Jm * Jm * Jm; dpen = J < 0 ? -4.0 * Jm * Jm * Jm : 0.0; ddpen = J < 0 ? 12.0 * J * J : 0.0; }
        cof3(f, C);
        for (i = 0; i < 9; i++) vJ += C[i] * C[i] * (double)vF[9 * e + i];
        {   /* the six triple products of det, each with its own rounding */
            const double a = f[2] * f[4] * f[6], b = f[1] * f[5] * f[6], c = f[2] * f[3] * f[7], dd = f[0] * f[5] * f[7],
                         ee = f[1] * f[3] * f[8], ff = f[0] * f[4] * f[8];
            vJ += u2 * (a * a + b * b + c * c + dd * dd + ee * ee + ff * ff);
        }
        for (i = 0; i < 3; i++)               /* cofactor entries C_ij = F_a F_d - F_b F_c */
            for (j = 0; j < 3; j++) {
                const int r0 = i == 0 ? 1 : 0, r1 = i == 2 ? 1 : 2, c0 = j == 0 ? 1 : 0, c1i = j == 2 ? 1 : 2;
                const double Fa = f[3 * r0 + c0], Fd = f[3 * r1 + c1i], Fb = f[3 * r0 + c1i], Fc = f[3 * r1 + c0];
                const double va = vF[9 * e + 3 * r0 + c0], vdd = vF[9 * e + 3 * r1 + c1i], vb = vF[9 * e + 3 * r0 + c1i],
                             vc = vF[9 * e + 3 * r1 + c0];
                vC[3 * i + j] = Fd * Fd * va + Fa * Fa * vdd + Fc * Fc * vb + Fb * Fb * vc + u2 * (Fa * Fd * Fa * Fd + Fb * Fc * Fb * Fc);
            }
        for (i = 0; i < 9; i++) {
            const double vPb = ddpen * ddpen * vJ * C[i] * C[i] + dpen * dpen * vC[i] + u2 * dpen * C[i] * dpen * C[i];
            Pm[i] = c1 * Q[i] + c2 * dpen * C[i];
            vP[i] = c1 * c1 * vQ[i] + c2 * c2 * vPb + u2 * Pm[i] * Pm[i];
        }
        for (i = 0; i < 3; i++)
            for (k = 0; k < 3; k++) {
                double s = 0.0, v = 0.0;
                for (j = 0; j < 3; j++) {
                    const double w = di[3 * k + j];
                    s += Pm[3 * i + j] * w;
                    v += (vP[3 * i + j] + u2 * Pm[3 * i + j] * Pm[3 * i + j]) * w * w;
                }
                d[3 * i + k] = s;
                vd[3 * i + k] = v;
            }
        {
            double all_vd = 0.0, all_d2 = 0.0, f02 = 0.0;
            for (k = 0; k < 3; k++) {
                double sv = 0.0, sd2 = 0.0;
                for (i = 0; i < 3; i++) { sv += vd[3 * i + k]; sd2 += d[3 * i + k] * d[3 * i + k]; }
#pragma omp atomic
                out_std_g[t[k + 1]] += sv + u2 * sd2;
                all_vd += sv;
                all_d2 += sd2;
            }
            for (i = 0; i < 3; i++) {
                const double f0 = -(d[3 * i] + d[3 * i + 1] + d[3 * i + 2]);
                f02 += f0 * f0;
            }
#pragma omp atomic
            out_std_g[t[0]] += all_vd + u2 * f02 + u2 * all_d2;
        }
        Eb += pen;
        sdpvJ += dpen * dpen * vJ;
        spen2 += pen * pen;
    }
    for (i = 0; i < n; i++) out_std_g[i] = sqrt(out_std_g[i]);
    {
        const double part = m > 1024 ? 1024.0 / (double)m : 1.0;
        const double vE = c1 * c1 * (sHHvH + 4 * u2 * Es * Es * part + u2 * 0.25 * sH4)
                        + c2 * c2 * (sdpvJ + u2 * spen2 + 4 * u2 * Eb * Eb * part);
        *out_std_E = sqrt(vE) + c1 * 0.5 * svH;
    }
done:
    free(dminv); free(F); free(H); free(vF); free(vH); free(nbr);
    return rc;
}
