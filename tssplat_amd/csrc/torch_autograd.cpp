// The autograd nodes of the energy as C++ torch::autograd::Functions (Python module tssplat_amd._tsamd_autograd).
//
// What they stand for: SmoothnessBarrierFunc of /root/reference/energies/smooth_barrier.py:9-31 -- forward = the energy,
// backward = grad_output x dE/dx -- for code shaped like /root/reference/trainer.py:94-130
// (loss = image_loss + energy(x, it, c1, c2); loss.backward()).  The Python twins (energies/smooth_barrier.py:
// SmoothnessBarrierFunc, energies/graphed.py: GraphReplayFunc) cost 70-110 us of host time per step on the MI355X box
// for 17 us of kernels on a 64-sphere batch (profiles/r03_host_overhead.txt): a Python autograd.Function round trip,
// ctypes marshalling of ten arguments, two Python frames per pass.  Here the node is created and run by the C++ autograd
// engine and the evaluation is ONE call into the C ABI (include/tssplat_amd.h):
//     energy_replay -> tsamd_graph_launch   (the library-owned HIP graph; static energy / gradient buffers)
//     energy_eval   -> tsamd_forward_backward (eager; fresh tensors)
// The library is not linked: tssplat_amd/_capi.py hands over the addresses of the entry points it has already loaded
// (set_entry_points), so there is one libtssplat_amd.so in the process and this file needs no HIP code of its own --
// torch only supplies tensors, the current stream and the autograd graph.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

namespace {

using launch_fn = int (*)(void *graph, float c1, float c2, void *stream);
using fwdbwd_fn = int (*)(void *handle, const float *x, const float *grad_out, float c1, float c2, int order, void *stream, float *energy,
                          float *grad);
using err_fn = const char *(*)();

launch_fn g_graph_launch = nullptr;
fwdbwd_fn g_forward_backward = nullptr;
err_fn g_last_error = nullptr;

void set_entry_points(int64_t graph_launch, int64_t forward_backward, int64_t last_error)
{
    g_graph_launch = reinterpret_cast<launch_fn>(graph_launch);
    g_forward_backward = reinterpret_cast<fwdbwd_fn>(forward_backward);
    g_last_error = reinterpret_cast<err_fn>(last_error);
}

void *current_stream(const at::Tensor &t) { return static_cast<void *>(c10::hip::getCurrentHIPStream(t.device().index()).stream()); }

void check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, "tssplat_amd: ", what, " failed (status ", rc, "): ", g_last_error ? g_last_error() : "");
}

// grad_output may arrive on the CPU (a 0-dim constant) or in another dtype: bring it to the gradient's device as float32
at::Tensor as_scale(const at::Tensor &go, const at::Tensor &like)
{
    if (go.device() == like.device() && go.scalar_type() == at::kFloat) return go;
    return go.detach().to(like.device(), at::kFloat, /*non_blocking=*/true);
}

// ---- HIP-graph replay behind an autograd node (energies/graphed.py: GraphedSmoothnessBarrier owns graph and buffers) ----
struct ReplayFunction : public torch::autograd::Function<ReplayFunction> {
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &x, int64_t graph, double c1, double c2,
                              const at::Tensor &energy, const at::Tensor &grad, const at::Tensor &ticket)
    {
        TORCH_CHECK(g_graph_launch, "tssplat_amd._tsamd_autograd: entry points not set");
        check(g_graph_launch(reinterpret_cast<void *>(graph), float(c1), float(c2), current_stream(x)), "tsamd_graph_launch");
        int64_t *t = ticket.data_ptr<int64_t>();   // (a CPU counter shared with the Python owner of the buffers)
        ctx->saved_data["ticket_value"] = ++*t;
        ctx->saved_data["ticket"] = ticket;
        ctx->saved_data["grad"] = grad;
        return energy.detach();   // a fresh tensor object on the static buffer: valid until the next evaluation
    }

    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::tensor_list grad_outputs)
    {
        const at::Tensor ticket = ctx->saved_data["ticket"].toTensor();
        TORCH_CHECK(ctx->saved_data["ticket_value"].toInt() == *ticket.data_ptr<int64_t>(),
                    "backward() of a graph-replayed energy after a newer evaluation overwrote its static gradient buffer: call backward() "
                    "before the next forward, or use graph=False");
        const at::Tensor grad = ctx->saved_data["grad"].toTensor();
        at::Tensor gx;
        if (grad_outputs[0].defined()) gx = grad * as_scale(grad_outputs[0], grad);   // a fresh tensor: x.grad never aliases the buffer
        return {gx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor energy_replay(const at::Tensor &x, int64_t graph, double c1, double c2, const at::Tensor &energy, const at::Tensor &grad,
                         const at::Tensor &ticket)
{
    return ReplayFunction::apply(x, graph, c1, c2, energy, grad, ticket);
}

// ---- eager evaluation: one fused forward+backward launch in forward, grad_output x gradient in backward ----
struct EvalFunction : public torch::autograd::Function<EvalFunction> {
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &x, int64_t handle, double c1, double c2, int64_t order)
    {
        TORCH_CHECK(g_forward_backward, "tssplat_amd._tsamd_autograd: entry points not set");
        TORCH_CHECK(x.is_cuda() && x.scalar_type() == at::kFloat, "tssplat_amd: x must be a float32 GPU tensor");
        const at::Tensor xc = x.contiguous();
        at::Tensor energy = at::empty({}, xc.options());
        at::Tensor grad = at::empty_like(xc);
        check(g_forward_backward(reinterpret_cast<void *>(handle), xc.data_ptr<float>(), nullptr, float(c1), float(c2), int(order),
                                 current_stream(xc), energy.data_ptr<float>(), grad.data_ptr<float>()),
              "tsamd_forward_backward");
        ctx->saved_data["grad"] = grad;
        return energy;
    }

    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::tensor_list grad_outputs)
    {
        const at::Tensor grad = ctx->saved_data["grad"].toTensor();
        at::Tensor gx;
        if (grad_outputs[0].defined()) gx = grad * as_scale(grad_outputs[0], grad);
        return {gx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor energy_eval(const at::Tensor &x, int64_t handle, double c1, double c2, int64_t order)
{
    return EvalFunction::apply(x, handle, c1, c2, order);
}

}  // namespace

void bind_energy_exchange(py::module &m);   // torch_exchange.cpp: the multi-GPU energy exchange's helper thread

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    bind_energy_exchange(m);
    m.def("set_entry_points", &set_entry_points, "addresses of tsamd_graph_launch, tsamd_forward_backward, tsamd_last_error");
    m.def("energy_replay", &energy_replay, "HIP-graph replay of the fused evaluation behind a C++ autograd node");
    m.def("energy_eval", &energy_eval, "eager fused evaluation behind a C++ autograd node");
}
