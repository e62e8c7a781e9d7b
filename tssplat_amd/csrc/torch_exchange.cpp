// The energy exchange of the sharded path with its helper thread in C++ (Python class tssplat_amd._tsamd_autograd.EnergyExchange;
// host code, no device code of its own).
//
// Independent tet-spheres shard over the GPUs of a node by whole spheres (the reference concatenates them with a running vertex
// offset, /root/reference/geometry/tetmesh_geometry.py:310-331); the only cross-rank step of an evaluation is the sum of the
// scalar energies.  tssplat_amd/sharding.py: OverlappedEnergyAllReduce issues that all-reduce once per evaluation from a helper
// thread, so that the training thread never pays for a collective call.  A Python helper thread still competes for the
// interpreter lock: measured on the MI355X box (tools/host_overhead.py, 64 x kuhn8) reserve + commit cost the training thread 45 us
// per step, of a 64 us step -- the hand-offs of the lock around every call of the helper.  This class is the same protocol with a
// std::thread: reserve() / commit() are a few hundred nanoseconds of bookkeeping plus one event record, the collective is issued
// through c10d::ProcessGroup::allreduce without the interpreter, value() releases the lock while it waits.
//
//   reserve()        -> (ticket, one-element view of the ring): the caller has the local energy written there on the current stream
//   commit(ticket)   -> records an event on the current stream, queues the slot (every > 1: the `every` slots of a window, with its
//                       last ticket -- one collective per window, for steps so short that even an asynchronous collective per
//                       step shows on the GPU's timeline; a value is then readable once its window has been issued)
//   helper thread    -> side stream waits for the event; pg->allreduce(slot) on the side stream (RCCL's own stream orders itself
//                       behind it); stores the work handle
//   value(ticket)    -> waits until issued, work->wait() (stream-side for RCCL), returns a clone of the slot
#include <torch/extension.h>

#include <ATen/hip/HIPEvent.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

namespace {

class EnergyExchange {
   public:
    EnergyExchange(const at::Tensor &ring, c10::optional<c10::intrusive_ptr<c10d::ProcessGroup>> pg, int64_t every)
        : ring_(ring), depth_(ring.numel()), every_(every), cuda_(ring.is_cuda())
    {
        TORCH_CHECK(ring.dim() == 1 && ring.scalar_type() == at::kFloat && ring.is_contiguous() && depth_ >= 2,
                    "EnergyExchange: the ring must be a contiguous 1-D float32 tensor of at least two slots");
        TORCH_CHECK(every >= 1 && depth_ % every == 0 && depth_ >= 2 * every, "EnergyExchange: `every` must divide the ring depth at least twice");
        if (pg.has_value()) pg_ = *pg;
        for (int64_t s = 0; s < depth_; ++s) slots_.push_back(ring_.narrow(0, s, 1));
        ticket_of_slot_.assign(size_t(depth_), -1);
        issued_.assign(size_t(depth_), 1);
        works_.resize(size_t(depth_));
        if (cuda_) {
            events_ = std::make_unique<at::cuda::CUDAEvent[]>(size_t(depth_));
            side_ = std::make_unique<c10::hip::HIPStream>(c10::hip::getStreamFromPool(/*isHighPriority=*/false, ring_.device().index()));
        }
    }

    ~EnergyExchange() { close(); }

    std::tuple<int64_t, at::Tensor> reserve()
    {
        const int64_t t = next_++, s = t % depth_;
        if (ticket_of_slot_[size_t(s)] >= 0) settle(s);   // depth tickets old: long done; orders the overwrite behind it
        ticket_of_slot_[size_t(s)] = t;
        return {t, slots_[size_t(s)]};
    }

    // Tickets are committed in order.  With every > 1 the collective covers the `every` slots of a window and is issued when the
    // window's LAST ticket is committed (the ring depth is a multiple of `every`: a window never wraps); flush() issues a partial one.
    void commit(int64_t ticket)
    {
        TORCH_CHECK(ticket == committed_ && ticket < next_, "EnergyExchange.commit: tickets are committed in the order they were reserved");
        ++committed_;
        if (committed_ % every_ == 0) issue(issued_upto_, committed_ - issued_upto_);   // (what a flush() has not sent already)
    }

    // the all-reduce of the committed part of the current window now (collective: every rank, at the same ticket)
    void flush()
    {
        if (committed_ > issued_upto_) issue(issued_upto_, committed_ - issued_upto_);
    }

    at::Tensor value(int64_t ticket)
    {
        TORCH_CHECK(ticket >= 0 && ticket < next_, "EnergyExchange.value: unknown ticket ", ticket);
        const int64_t s = ticket % depth_;
        TORCH_CHECK(ticket_of_slot_[size_t(s)] == ticket, "the job-wide energy of evaluation ", ticket, " has expired: ", next_ - ticket,
                    " evaluations ago, the ring keeps ", depth_, " (read it sooner, or build the module with a larger `depth`)");
        TORCH_CHECK(ticket < issued_upto_, "the job-wide energy of evaluation ", ticket, " is not on its way yet: with every = ", every_,
                    " the all-reduce of a window is issued with its last evaluation (", issued_upto_, " evaluations are covered so far).  Read it "
                    "later, call flush() on EVERY rank, or build the module with every = 1");
        settle(s);
        return slots_[size_t(s)].squeeze(0).clone();
    }

    void drain()
    {
        for (int64_t s = 0; s < depth_; ++s)
            if (ticket_of_slot_[size_t(s)] >= 0 && ticket_of_slot_[size_t(s)] < issued_upto_) settle(s);
    }

    void close()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_q_.notify_all();
        if (thread_.joinable()) {
            py::gil_scoped_release nogil;   // (the worker never takes the interpreter lock, but a collective may take a while)
            thread_.join();
        }
    }

    int64_t collectives()
    {
        std::lock_guard<std::mutex> lock(m_);
        return collectives_;
    }
    int64_t depth() const { return depth_; }
    int64_t issued_tickets() const { return next_; }

   private:
    // one collective over the slots of tickets [first, first + count)
    void issue(int64_t first, int64_t count)
    {
        issued_upto_ = first + count;
        if (!pg_) return;                                  // no process group: the exchange is the identity (issued_ stays 1)
        const int64_t s0 = first % depth_;
        if (cuda_) events_[size_t(s0)].record(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(ring_.device().index()));
        {
            std::lock_guard<std::mutex> lock(m_);
            for (int64_t k = 0; k < count; ++k) issued_[size_t(s0 + k)] = 0;
            q_.push_back({s0, count});
            if (!thread_.joinable()) thread_ = std::thread([this] { worker(); });
        }
        cv_q_.notify_one();
    }

    // the collective that last used slot s has been issued and the CALLING thread's current stream is ordered behind its completion
    void settle(int64_t s)
    {
        c10::intrusive_ptr<c10d::Work> w;
        {
            py::gil_scoped_release nogil;
            std::unique_lock<std::mutex> lock(m_);
            cv_done_.wait(lock, [&] { return issued_[size_t(s)] != 0; });
            TORCH_CHECK(error_.empty(), "the energy all-reduce failed in the helper thread: ", error_);
            w = works_[size_t(s)];
            // every slot of a window shares one handle: one stream-side wait serves them all (a wait per SLOT put a wait packet on the
            // training stream every step: +4.5 us per step at 64 x kuhn19, tools/module_breakdown.py)
            const int64_t w0 = s - s % every_;
            for (int64_t k = w0; k < w0 + every_; ++k)
                if (works_[size_t(k)] == w) works_[size_t(k)].reset();
        }
        if (w && !w->isCompleted()) {   // (a finished collective needs no wait packet on the calling stream: slot re-use, 256 tickets later)
            py::gil_scoped_release nogil;
            w->wait();   // RCCL: the current stream waits; gloo: the host does
        }
    }

    void worker()
    {
        c10d::AllreduceOptions opts;
        opts.reduceOp = c10d::ReduceOp::SUM;
        for (;;) {
            int64_t s, count;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_q_.wait(lock, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;   // (stop: everything queued has been issued)
                s = q_.front().first;
                count = q_.front().second;
                q_.pop_front();
            }
            c10::intrusive_ptr<c10d::Work> w;
            std::string err;
            try {
                std::vector<at::Tensor> t{count == 1 ? slots_[size_t(s)] : ring_.narrow(0, s, count)};
                if (cuda_) {
                    c10::hip::HIPStreamGuard on_side(*side_);
                    events_[size_t(s)].block(c10::hip::HIPStreamMasqueradingAsCUDA(*side_));
                    w = pg_->allreduce(t, opts);
                } else {
                    w = pg_->allreduce(t, opts);
                }
            } catch (const std::exception &e) {
                err = e.what();
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                for (int64_t k = 0; k < count; ++k) {   // (every slot of the window waits on the same handle)
                    works_[size_t(s + k)] = w;
                    issued_[size_t(s + k)] = 1;
                }
                if (!err.empty() && error_.empty()) error_ = err;
                ++collectives_;
            }
            cv_done_.notify_all();
        }
    }

    at::Tensor ring_;
    int64_t depth_, every_;
    bool cuda_;
    c10::intrusive_ptr<c10d::ProcessGroup> pg_;
    std::vector<at::Tensor> slots_;
    std::unique_ptr<at::cuda::CUDAEvent[]> events_;
    std::unique_ptr<c10::hip::HIPStream> side_;
    std::vector<int64_t> ticket_of_slot_;
    int64_t next_ = 0, committed_ = 0, issued_upto_ = 0;
    // shared with the helper thread (m_)
    std::mutex m_;
    std::condition_variable cv_q_, cv_done_;
    std::deque<std::pair<int64_t, int64_t>> q_;   // (first slot, slots)
    std::vector<char> issued_;
    std::vector<c10::intrusive_ptr<c10d::Work>> works_;
    std::string error_;
    int64_t collectives_ = 0;
    bool stop_ = false;
    std::thread thread_;
};

}  // namespace

void bind_energy_exchange(py::module &m)
{
    py::class_<EnergyExchange, std::shared_ptr<EnergyExchange>>(m, "EnergyExchange")
        .def(py::init<const at::Tensor &, c10::optional<c10::intrusive_ptr<c10d::ProcessGroup>>, int64_t>(), py::arg("ring"), py::arg("group"),
             py::arg("every") = 1)
        .def("flush", &EnergyExchange::flush)
        .def("reserve", &EnergyExchange::reserve)
        .def("commit", &EnergyExchange::commit)
        .def("value", &EnergyExchange::value)
        .def("drain", &EnergyExchange::drain)
        .def("close", &EnergyExchange::close)
        .def_property_readonly("collectives", &EnergyExchange::collectives)
        .def_property_readonly("depth", &EnergyExchange::depth)
        .def_property_readonly("issued_tickets", &EnergyExchange::issued_tickets);
}
