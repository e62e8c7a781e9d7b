// Host harness around lines 9-102 of the reference's tet_spheres_cuda.cu (extracted by the Makefile next to this
// file into oracle/_ref/ref_det_extract.inc) -- TEST INFRASTRUCTURE, our code; the included lines are the reference's.
//
// The extract holds: elt / det / ddetA_dA (templates) and the kernels cuda_forward_det / cuda_backward_det.  The CUDA
// keywords are defined away and the kernels' thread coordinates become three globals that the C entry points below
// set before calling a kernel body once per element -- so the reference's own arithmetic runs, on the host.
#include <cmath>
#include <cstdint>

#define __host__
#define __device__
#define __global__
namespace {
struct Idx3 {
    int x, y, z;
};
Idx3 blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, threadIdx = {0, 0, 0};
}  // namespace

#include "ref_det_extract.inc"

extern "C" {

// det / ddetA_dA instantiated at float (what the kernels use) and double
float ref_det_f32(const float *M) { return det<float>(M); }
double ref_det_f64(const double *M) { return det<double>(M); }
void ref_cof_f32(const float *A, float *out) { ddetA_dA<float>(A, out); }
void ref_cof_f64(const double *A, double *out) { ddetA_dA<double>(A, out); }

// the two penalty kernels, element by element (gidx = blockIdx.x with 1x1 blocks)
void ref_forward_det(int nele, float *tetF, float *tetJ, int order)
{
    blockDim = {1, 1, 1};
    threadIdx = {0, 0, 0};
    for (int e = 0; e < nele; ++e) {
        blockIdx.x = e;
        cuda_forward_det(nele, tetF, tetJ, order);
    }
}

void ref_backward_det(int nele, float *tetF, float *tet_dJ_dF, int order)
{
    blockDim = {1, 1, 1};
    threadIdx = {0, 0, 0};
    for (int e = 0; e < nele; ++e) {
        blockIdx.x = e;
        cuda_backward_det(nele, tetF, tet_dJ_dF, order);
    }
}

}  // extern "C"
