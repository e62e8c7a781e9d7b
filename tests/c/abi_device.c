/* Plain-C consumer of the DEVICE entry points of include/tssplat_amd.h: a compiled C99 program (no Python, no torch,
 * no ctypes) creates a handle on the GPU, evaluates energy + gradient through tsamd_forward_backward,
 * tsamd_forward, tsamd_backward, tsamd_evaluate_dev_coef and tsamd_graph_create / tsamd_graph_launch / tsamd_graph_launch_to, and checks them against the float64 C oracle
 * (oracle/c/tet_energy_oracle.c, test infrastructure) on a small lattice of tets.
 * Built and run by tests/test_gpu_parity.py::test_compiled_c_consumer_on_device:
 *   gcc -std=c99 abi_device.c -ltssplat_amd -ltet_energy_oracle -lamdhip64 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "tssplat_amd.h"

int tso_energy_grad(int64_t n, int64_t m, const float *rest, const int32_t *tets, const int32_t *nbr_in, const float *x,
                    float c1f, float c2f, int order, double grad_output, double *E_out, double *grad);

#define K 6 /* (K+1)^3 lattice vertices, 6 K^3 tets (Kuhn split of every cube) */
#define NV ((K + 1) * (K + 1) * (K + 1))
#define NT (6 * K * K * K)

static int vid(int i, int j, int l) { return (i * (K + 1) + j) * (K + 1) + l; }

#define HIPCHK(call)                                                            \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));          \
            return 20;                                                          \
        }                                                                       \
    } while (0)

int main(void)
{
    static float rest[3 * NV], x[3 * NV], g_gpu[3 * NV];
    static int32_t tets[4 * NT];
    static double g_ref[3 * NV];
    static const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    static const int odd[6] = {0, 1, 1, 0, 0, 1};
    int i, j, l, p, t = 0, order, rc;
    unsigned s = 12345u;
    float *d_x = NULL, *d_g = NULL, *d_e = NULL, *d_coef = NULL, *d_go = NULL;
    tsamd_options opt;
    tsamd_handle *h = NULL;
    const float c1 = 3e-4f, c2 = 2e-4f, go = 0.5f;
    const float coef[2] = {3e-4f, 2e-4f};

    for (i = 0; i <= K; ++i)
        for (j = 0; j <= K; ++j)
            for (l = 0; l <= K; ++l) {
                float *r = rest + 3 * vid(i, j, l);
                r[0] = 0.1f * (float)i + 0.013f * (float)((j * 7 + l * 3) % 5);
                r[1] = 0.1f * (float)j + 0.011f * (float)((i * 5 + l * 2) % 7);
                r[2] = 0.1f * (float)l + 0.012f * (float)((i * 3 + j * 11) % 3);
            }
    for (i = 0; i < K; ++i)
        for (j = 0; j < K; ++j)
            for (l = 0; l < K; ++l)
                for (p = 0; p < 6; ++p) {
                    int c[3], a;
                    int32_t *tt = tets + 4 * t++;
                    c[0] = i, c[1] = j, c[2] = l;
                    tt[0] = vid(c[0], c[1], c[2]);
                    for (a = 0; a < 3; ++a) {
                        c[perm[p][a]] += 1;
                        tt[a + 1] = vid(c[0], c[1], c[2]);
                    }
                    if (odd[p]) { int32_t w = tt[1]; tt[1] = tt[2]; tt[2] = w; }   /* keep det Dm > 0 */
                }
    for (i = 0; i < 3 * NV; ++i) {   /* deformation large enough to invert some tets */
        s = s * 1664525u + 1013904223u;
        x[i] = rest[i] + 0.06f * ((float)(s >> 8) / 16777216.0f - 0.5f);
    }

    memset(&opt, 0, sizeof opt);
    opt.struct_size = (int32_t)sizeof opt;
    opt.abi_version = TSAMD_ABI_VERSION;
    opt.device = 0;
    opt.lds_budget_bytes = 24000; /* several tiles: halo slots, staged vertices and the finish kernel take part */
    if (tsamd_create(rest, NV, tets, NT, &opt, &h) != TSAMD_OK) {
        fprintf(stderr, "create: %s\n", tsamd_last_error());
        return 11;
    }
    HIPCHK(hipMalloc((void **)&d_x, sizeof x));
    HIPCHK(hipMalloc((void **)&d_g, sizeof g_gpu));
    HIPCHK(hipMalloc((void **)&d_e, sizeof(float)));
    HIPCHK(hipMalloc((void **)&d_coef, sizeof coef));
    HIPCHK(hipMalloc((void **)&d_go, sizeof go));
    HIPCHK(hipMemcpy(d_x, x, sizeof x, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_coef, coef, sizeof coef, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_go, &go, sizeof go, hipMemcpyHostToDevice));

    for (order = 2; order <= 4; order += 2) {
        double E_ref[3], gn = 0.0, err;
        float e_gpu = 0.f, e2 = 0.f;
        int variant;
        if ((rc = tso_energy_grad(NV, NT, rest, tets, NULL, x, c1, c2, order, (double)go, E_ref, g_ref)) != 0) return 12;
        for (i = 0; i < 3 * NV; ++i) gn += g_ref[i] * g_ref[i];
        gn = sqrt(gn);
        if (E_ref[2] <= 0.0) return 13; /* the case is meant to exercise the inversion penalty */
        for (variant = 0; variant < 5; ++variant) {
            HIPCHK(hipMemset(d_g, 0xff, sizeof g_gpu));
            if (variant == 0)
                rc = tsamd_forward_backward(h, d_x, d_go, c1, c2, order, NULL, d_e, d_g);
            else if (variant == 1)
                rc = tsamd_evaluate_dev_coef(h, d_x, d_go, d_coef, order, NULL, d_e, d_g);
            else if (variant == 2)
                rc = tsamd_forward(h, d_x, c1, c2, order, NULL, d_e) || tsamd_backward(h, d_x, d_go, c1, c2, order, NULL, d_g);
            else if (variant == 3) { /* the library-owned HIP graph: first launch with other coefficients, then with the real ones */
                tsamd_graph *gr = NULL;
                rc = tsamd_graph_create(h, d_x, d_go, order, d_e, d_g, &gr) || tsamd_graph_launch(gr, 3.f * c1, 0.5f * c2, NULL) ||
                     tsamd_graph_launch(gr, c1, c2, NULL);
                HIPCHK(hipDeviceSynchronize());
                tsamd_graph_destroy(gr);
            } else { /* the replay with its outputs redirected per launch: energy ALSO to a second address, gradient to another buffer */
                tsamd_graph *gr = NULL;
                float *d_e2 = NULL, *d_g2 = NULL, e_copy = -1.f, e_own = -2.f;
                HIPCHK(hipMalloc((void **)&d_e2, sizeof(float)));
                HIPCHK(hipMalloc((void **)&d_g2, sizeof g_gpu));
                HIPCHK(hipMemset(d_g2, 0xff, sizeof g_gpu));
                rc = tsamd_graph_create(h, d_x, d_go, order, d_e, d_g, &gr) || tsamd_graph_launch(gr, c1, c2, NULL) /* own buffers first */ ||
                     tsamd_graph_launch_to(gr, c1, c2, NULL, d_e2, d_g2);
                HIPCHK(hipDeviceSynchronize());
                HIPCHK(hipMemcpy(&e_copy, d_e2, sizeof e_copy, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(&e_own, d_e, sizeof e_own, hipMemcpyDeviceToHost));
                if (rc == TSAMD_OK && e_copy != e_own) return 18;              /* both destinations hold the same value */
                HIPCHK(hipMemcpy(d_g, d_g2, sizeof g_gpu, hipMemcpyDeviceToDevice)); /* (checked below like every variant's gradient) */
                /* a graph without a gradient cannot redirect one: loud */
                if (rc == TSAMD_OK) {
                    tsamd_graph *ge = NULL;
                    if (tsamd_graph_create(h, d_x, NULL, order, d_e, NULL, &ge) != TSAMD_OK) return 19;
                    if (tsamd_graph_launch_to(ge, c1, c2, NULL, NULL, d_g2) == TSAMD_OK) return 20;
                    tsamd_graph_destroy(ge);
                }
                tsamd_graph_destroy(gr);
                hipFree(d_e2), hipFree(d_g2);
            }
            if (rc != TSAMD_OK) {
                fprintf(stderr, "evaluate (variant %d): %s\n", variant, tsamd_last_error());
                return 14;
            }
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(&e_gpu, d_e, sizeof e_gpu, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(g_gpu, d_g, sizeof g_gpu, hipMemcpyDeviceToHost));
            err = 0.0;
            for (i = 0; i < 3 * NV; ++i) err += ((double)g_gpu[i] - g_ref[i]) * ((double)g_gpu[i] - g_ref[i]);
            err = sqrt(err);
            printf("order %d variant %d: E %.7g (oracle %.7g)  |dg| %.3e  |g| %.3e\n", order, variant, (double)e_gpu, E_ref[0], err, gn);
            if (fabs((double)e_gpu - E_ref[0]) > 2e-6 * fabs(E_ref[0])) return 15;
            if (err > 5e-6 * gn) return 16;
            if (variant == 0) e2 = e_gpu;
            else if (e_gpu != e2) return 17; /* same arithmetic whichever entry point: bitwise equal energy */
        }
    }
    hipFree(d_x), hipFree(d_g), hipFree(d_e), hipFree(d_coef), hipFree(d_go);
    tsamd_destroy(h);
    printf("abi device ok\n");
    return 0;
}
