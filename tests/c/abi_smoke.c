/* Plain-C consumer of include/tssplat_amd.h: proves the boundary is a C ABI (C99, no C++ types), that the
 * header compiles on its own, and that the host-only entry points work without a GPU.
 * Built and run by tests/test_c_abi.py:  gcc -std=c99 -pedantic -Wall -Werror abi_smoke.c -ltssplat_amd */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tssplat_amd.h"

int main(void)
{
    /* two tets sharing the face (1,2,3) */
    const float rest[15] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 1};
    const int32_t tets[8] = {0, 1, 2, 3, 4, 1, 3, 2};
    tsamd_options opt;
    tsamd_handle *h = NULL;
    tsamd_plan_info info;
    const int32_t *adj = NULL;
    int32_t vid[8], faces[3 * 8];
    int64_t nv = 0, nf = 0;

    if (strstr(tsamd_version(), "gfx950") == NULL) return 10;
    memset(&opt, 0, sizeof opt);
    opt.struct_size = (int32_t)sizeof opt;
    opt.abi_version = TSAMD_ABI_VERSION;
    opt.device = -1;
    opt.host_only = 1; /* plan only: no HIP call is made */
    if (tsamd_create(rest, 5, tets, 2, &opt, &h) != TSAMD_OK) {
        fprintf(stderr, "create: %s\n", tsamd_last_error());
        return 11;
    }
    if (tsamd_num_vertices(h) != 5 || tsamd_num_tets(h) != 2) return 12;
    if (tsamd_get_plan_info(h, &info) != TSAMD_OK || info.n_tiles != 1 || info.total_slots != 2) return 13;
    if (tsamd_get_adjacency(h, &adj) != TSAMD_OK) return 14;
    /* nbr[4e+k] = tet across the face opposite local vertex k: both tets see each other across vertex 0 */
    if (adj[0] != 1 || adj[4] != 0 || adj[1] != -1) return 15;
    /* a device entry point on a host-only handle fails loudly with its own status code */
    if (tsamd_forward(h, rest, 1.f, 1.f, 2, NULL, (float *)rest) != TSAMD_ERR_HOST_ONLY) return 16;
    tsamd_destroy(h);

    /* boundary extraction (host): 6 triangles over 5 vertices; sizes first, then the data */
    if (tsamd_extract_surface(tets, 2, 5, NULL, &nv, NULL, &nf) != TSAMD_OK || nv != 5 || nf != 6) return 17;
    if (tsamd_extract_surface(tets, 2, 5, vid, &nv, faces, &nf) != TSAMD_OK) return 18;
    if (vid[0] != 0 || vid[4] != 4) return 19;

    /* errors: status code + text */
    if (tsamd_create(rest, 5, tets, -1, &opt, &h) == TSAMD_OK) return 20;
    if (strlen(tsamd_last_error()) == 0) return 21;
    puts("abi smoke ok");
    return 0;
}
