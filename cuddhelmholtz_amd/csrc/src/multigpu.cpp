// One process, `world` devices: DDH with subdomains sharded over the devices and RCCL neighbour exchange (multigpu.hpp).
#include "cuddh/multigpu.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: the library is bound at run time, see Rccl below

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>

#include "cuddh/basis.hpp"
#include "cuddh/ddh.hpp"
#include "cuddh/helmholtz.hpp"
#include "cuddh/meshio.hpp"
#include "cuddh/partition.hpp"
#include "cuddh/launch.hpp"
#include "cuddh/mesh.hpp"
#include "cuddh/spaces.hpp"
#include "cuddh_hip.h"

namespace cuddh
{
    std::vector<int> TraceExchangePlan::rank_grid(int ndx, int ndy, int gx, int gy)
    {
        if (gx < 1 || gy < 1 || gx > ndx || gy > ndy)
            cuddh_error("TraceExchangePlan error: the rank grid does not fit the block grid.");
        std::vector<int> out(static_cast<std::size_t>(ndx) * ndy);
        for (int by = 0; by < ndy; ++by)
            for (int bx = 0; bx < ndx; ++bx)
                out[bx + static_cast<std::size_t>(ndx) * by] =
                    static_cast<int>((static_cast<long long>(bx) * gx) / ndx) + gx * static_cast<int>((static_cast<long long>(by) * gy) / ndy);
        return out;
    }

    TraceExchangePlan TraceExchangePlan::build(const int *B, int n_domains, int mx_fdof, int n_lambda, int rank, int world,
                                               const int *given_dom_rank)
    {
        if (rank < 0 || rank >= world)
            cuddh_error("TraceExchangePlan error: rank out of range.");
        TraceExchangePlan p;
        p.rank = rank;
        p.world = world;
        p.n_lambda = n_lambda;
        shard_range(n_domains, rank, world, p.dom_begin, p.dom_end);

        std::vector<int> dom_rank(n_domains);
        if (given_dom_rank)
        {
            for (int s = 0; s < n_domains; ++s)
            {
                if (given_dom_rank[s] < 0 || given_dom_rank[s] >= world)
                    cuddh_error("TraceExchangePlan error: dom_rank needs one rank in [0, world) per subdomain.");
                dom_rank[s] = given_dom_rank[s];
            }
        }
        else
            for (int r = 0; r < world; ++r)
            {
                int a, b;
                shard_range(n_domains, r, world, a, b);
                std::fill(dom_rank.begin() + a, dom_rank.begin() + b, r);
            }
        for (int s = 0; s < n_domains; ++s)
            if (dom_rank[s] == rank)
                p.domains.push_back(s);
        if (given_dom_rank) // [dom_begin, dom_end) only spans this rank's subdomains; use `domains`
        {
            p.dom_begin = p.domains.empty() ? 0 : p.domains.front();
            p.dom_end = p.domains.empty() ? 0 : p.domains.back() + 1;
        }
        std::vector<int> reader(n_lambda, -1), writer(n_lambda, -1);
        for (int s = 0; s < n_domains; ++s)
            for (int col = 0; col < 2; ++col)
                for (int i = 0; i < mx_fdof; ++i)
                {
                    const int t = B[i + static_cast<std::size_t>(mx_fdof) * (col + 2 * static_cast<std::size_t>(s))];
                    if (t < 0)
                        continue;
                    std::vector<int> &who = col == 0 ? reader : writer;
                    if (t >= n_lambda || who[t] >= 0)
                        cuddh_error("DDH slot table: a slot is used by two subdomains or is out of range");
                    who[t] = s;
                }
        for (int t = 0; t < n_lambda; ++t)
        {
            const int owner_dom = reader[t] >= 0 ? reader[t] : writer[t];
            const int owner = owner_dom >= 0 ? dom_rank[owner_dom] : -1;
            const int wrank = writer[t] >= 0 ? dom_rank[writer[t]] : -1;
            if (owner == rank)
            {
                p.owned.push_back(t);
                if (wrank >= 0 && wrank != rank)
                    p.recv[wrank].push_back(t);
            }
            else if (owner >= 0 && wrank == rank)
                p.send[owner].push_back(t);
        }
        // boundary subdomains = writers of the slots that are sent
        std::vector<char> is_boundary(n_domains, 0);
        for (const auto &kv : p.send)
            for (const int t : kv.second)
                is_boundary[writer[t]] = 1;
        for (const int s : p.domains)
            (is_boundary[s] ? p.boundary : p.interior).push_back(s);
        if (!p.boundary.empty())
        {
            const std::size_t pad = std::min<std::size_t>((8 - p.boundary.size() % 8) % 8, p.interior.size());
            p.boundary.insert(p.boundary.end(), p.interior.begin(), p.interior.begin() + pad);
            p.interior.erase(p.interior.begin(), p.interior.begin() + pad);
            std::sort(p.boundary.begin(), p.boundary.end());
        }
        return p;
    }

    namespace
    {
        // RCCL is bound with dlopen at first use instead of being a link-time dependency: a host process may already hold a
        // copy (PyTorch ships its own librccl.so), and two copies of the library in one process corrupt each other's state at
        // exit.  dlopen by name returns the copy that is already loaded, if any; otherwise ROCm's.
        struct Rccl
        {
            decltype(&ncclCommInitAll) CommInitAll = nullptr;
            decltype(&ncclCommDestroy) CommDestroy = nullptr;
            decltype(&ncclCommAbort) CommAbort = nullptr;
            decltype(&ncclGroupStart) GroupStart = nullptr;
            decltype(&ncclGroupEnd) GroupEnd = nullptr;
            decltype(&ncclSend) Send = nullptr;
            decltype(&ncclRecv) Recv = nullptr;
            decltype(&ncclAllReduce) AllReduce = nullptr;
            decltype(&ncclGetErrorString) GetErrorString = nullptr;

            static const Rccl &get()
            {
                static const Rccl api = load();
                return api;
            }

        private:
            static Rccl load()
            {
                void *h = nullptr;
                for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"})
                {
                    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD); // a copy the process already holds
                    if (h)
                        break;
                }
                if (!h)
                    for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
                    {
                        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                        if (h)
                            break;
                    }
                if (!h)
                    throw std::runtime_error(std::string("RCCL is not available: ") + dlerror());
                Rccl a;
                auto bind = [&](auto &fn, const char *sym)
                {
                    fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, sym));
                    if (!fn)
                        throw std::runtime_error(std::string("RCCL symbol missing: ") + sym);
                };
                bind(a.CommInitAll, "ncclCommInitAll");
                bind(a.CommDestroy, "ncclCommDestroy");
                bind(a.CommAbort, "ncclCommAbort");
                bind(a.GroupStart, "ncclGroupStart");
                bind(a.GroupEnd, "ncclGroupEnd");
                bind(a.Send, "ncclSend");
                bind(a.Recv, "ncclRecv");
                bind(a.AllReduce, "ncclAllReduce");
                bind(a.GetErrorString, "ncclGetErrorString");
                return a;
            }
        };

        void check_nccl(ncclResult_t r, const char *what)
        {
            if (r != ncclSuccess)
                throw std::runtime_error(std::string(what) + ": " + Rccl::get().GetErrorString(r));
        }

        using clk = std::chrono::steady_clock;
        double since(clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); }

        // Loopback transport (transport == 2): the ranks are host threads that SHARE one device, messages are device-to-device
        // copies between their buffers and reductions are summed on the host in rank order.  It carries no performance claim;
        // it exists so that everything of the N > 1 path except the RCCL calls themselves -- exchange plan, pack / unpack,
        // partitioned Krylov vectors, the reduce hook, one stream per thread -- runs on a one-GPU box (tests).
        struct Loopback
        {
            explicit Loopback(int world) : n(world), box(static_cast<std::size_t>(world) * world, nullptr), vals(world) {}
            /// all ranks meet here -- or leave with an exception as soon as any rank has failed (abort()), so that one rank's error
            /// never leaves the others waiting for it
            void barrier()
            {
                std::unique_lock<std::mutex> lk(m);
                if (failed)
                    throw std::runtime_error("multi-GPU solve: another rank failed");
                const long gen = generation;
                if (++arrived == n)
                {
                    arrived = 0;
                    ++generation;
                    cv.notify_all();
                }
                else
                {
                    cv.wait(lk, [&] { return generation != gen || failed; });
                    if (generation == gen)
                        throw std::runtime_error("multi-GPU solve: another rank failed");
                }
            }
            void abort()
            {
                {
                    std::lock_guard<std::mutex> lk(m);
                    failed = true;
                }
                cv.notify_all();
            }
            bool failed = false;
            int n;
            std::vector<const void *> box; // box[from * n + to]: device pointer of the message from -> to
            std::vector<std::vector<double>> vals;
            std::mutex m;
            std::condition_variable cv;
            int arrived = 0;
            long generation = 0;
        };

        // Runs body(rank) for rank 0 on the caller's thread and for the other ranks on their own threads.  One rank's failure ends
        // the call on every rank: the loopback barrier is released with an error and the RCCL communicators are aborted
        // (ncclCommAbort makes pending and later collectives on them return), so no thread stays blocked in a collective; the
        // first rank's OWN error is re-thrown, not the "another rank failed" of the ranks it released.  The caller gets its launch
        // stream and current device back whatever happens.
        template <typename Body>
        void run_ranks(int world, Loopback &loop_state, std::vector<ncclComm_t> &comms, Body body)
        {
            struct CallerState
            {
                hipStream_t st = stream();
                int dev = 0;
                CallerState() { (void)hipGetDevice(&dev); }
                ~CallerState()
                {
                    set_stream(st);
                    (void)hipSetDevice(dev);
                }
            } caller_state;
            std::vector<std::exception_ptr> errors(world);
            std::atomic<bool> any_failed{false};
            std::mutex abort_once;
            auto abort_all = [&]()
            {
                std::lock_guard<std::mutex> lock(abort_once);
                if (any_failed.exchange(true))
                    return;
                loop_state.abort();
                for (auto &c : comms)
                    if (c)
                    {
                        (void)Rccl::get().CommAbort(c);
                        c = nullptr;
                    }
            };
            auto guarded = [&](int rank)
            {
                try
                {
                    body(rank);
                }
                catch (...)
                {
                    errors[rank] = std::current_exception();
                    abort_all();
                }
                set_stream(nullptr); // the rank's streams are gone; this thread launches on no dead stream
            };
            std::vector<std::thread> threads;
            for (int r = 1; r < world; ++r)
                threads.emplace_back(guarded, r);
            guarded(0);
            for (auto &t : threads)
                t.join();
            for (auto &c : comms)
                if (c)
                    (void)Rccl::get().CommDestroy(c);
            std::exception_ptr first;
            for (const auto &e : errors)
            {
                if (!e)
                    continue;
                bool secondary = false;
                try
                {
                    std::rethrow_exception(e);
                }
                catch (const std::exception &ex)
                {
                    secondary = std::strstr(ex.what(), "another rank failed") != nullptr;
                }
                catch (...)
                {
                }
                if (!secondary)
                {
                    first = e;
                    break;
                }
                if (!first)
                    first = e;
            }
            if (first)
                std::rethrow_exception(first);
        }

        std::vector<ncclComm_t> make_comms(int world, bool use_rccl)
        {
            std::vector<ncclComm_t> comms(world, nullptr);
            if (use_rccl)
            {
                std::vector<int> devs(world);
                for (int r = 0; r < world; ++r)
                    devs[r] = r;
                check_nccl(Rccl::get().CommInitAll(comms.data(), world, devs.data()), "ncclCommInitAll");
            }
            return comms;
        }

        // sum of `count` device scalars over the ranks, in place, ordered on `st`
        void all_reduce_on(Loopback *loop, ncclComm_t comm, hipStream_t st, int rank, int world, void *d, size_t count, ncclDataType_t type)
        {
            if (loop)
            {
                const size_t bytes = count * (type == ncclDouble ? 8 : 4);
                std::vector<char> mine(bytes);
                detail::check_hip(cuddh_hip_copy_d2h_on(mine.data(), d, bytes, st), "loopback reduce");
                detail::check_hip(cuddh_hip_stream_sync(st), "stream sync");
                std::vector<double> &slot = loop->vals[rank];
                slot.resize(count);
                for (size_t i = 0; i < count; ++i)
                    slot[i] = type == ncclDouble ? reinterpret_cast<const double *>(mine.data())[i] : reinterpret_cast<const float *>(mine.data())[i];
                loop->barrier();
                std::vector<double> sum(count, 0.0);
                for (int r = 0; r < world; ++r) // rank order: every rank gets the same bits
                    for (size_t i = 0; i < count; ++i)
                        sum[i] += loop->vals[r][i];
                loop->barrier(); // everybody has read the slots
                for (size_t i = 0; i < count; ++i)
                    if (type == ncclDouble)
                        reinterpret_cast<double *>(mine.data())[i] = sum[i];
                    else
                        reinterpret_cast<float *>(mine.data())[i] = static_cast<float>(sum[i]);
                detail::check_hip(cuddh_hip_copy_h2d_on(d, mine.data(), bytes, st), "loopback reduce");
                detail::check_hip(cuddh_hip_stream_sync(st), "stream sync");
                return;
            }
            if (comm)
                check_nccl(Rccl::get().AllReduce(d, d, count, type, ncclSum, comm, st), "ncclAllReduce");
        }

        // everything one device needs, built and used by that device's host thread only
        struct Rank
        {
            int rank, world, device;
            Loopback *loop = nullptr;  // test transport, see above
            ncclComm_t comm = nullptr; // null: no communicator (world == 1 without force_rccl)
            hipStream_t st = nullptr;
            hipStream_t st_side = nullptr; // split schedule: boundary subdomains and the exchange
            hipEvent_t ev_main = nullptr, ev_side = nullptr;
            HostDeviceArray<int> boundary_ids, interior_ids, all_ids; // all_ids: only when this rank's subdomains are not one range
            std::unique_ptr<Mesh2D> mesh;
            std::unique_ptr<Basis> basis;
            std::unique_ptr<H1Space> fem;
            std::unique_ptr<DDH> F;
            TraceExchangePlan plan;
            std::map<int, HostDeviceArray<int>> send_slots, recv_slots;
            std::map<int, HostDeviceArray<float>> sbuf, rbuf;

            Rank() = default;
            Rank(const Rank &) = delete;
            ~Rank() // also on the exception path: operators first (they launch on st), then the streams and events
            {
                F.reset();
                fem.reset();
                if (st_side)
                {
                    (void)hipStreamSynchronize(st_side);
                    (void)hipStreamDestroy(st_side);
                }
                if (ev_main)
                    (void)hipEventDestroy(ev_main);
                if (ev_side)
                    (void)hipEventDestroy(ev_side);
                if (st)
                {
                    (void)hipStreamSynchronize(st);
                    (void)hipStreamDestroy(st);
                }
            }

            void sync() const { detail::check_hip(cuddh_hip_stream_sync(st), "stream sync"); }

            /// out <- traces written by this rank's subdomains into slots it owns + traces received from the other ranks
            void traces(const double *f, const float *lambda, float *out)
            {
                const int n = F->size();
                detail::check_hip(cuddh_hip_memset_zero(out, sizeof(float) * n, st), "trace zero fill");
                const bool split = st_side && world > 1 && boundary_ids.size() > 0;
                hipStream_t xst = st; // the stream the exchange runs on
                if (!split && all_ids.size() > 0)
                    F->internals().solve_listed(all_ids.device_read(), all_ids.size(), f, lambda, out);
                else if (!split)
                    F->local_traces(plan.dom_begin, plan.dom_end, f, lambda, out);
                else
                {
                    // boundary subdomains: one listed launch with issue priority on the side stream, the exchange behind it;
                    // the interior on the main stream meanwhile (dist.py::NeighbourShardedDDH.traces, profiles/r02/overlap_timeline.txt)
                    const auto &core = F->internals();
                    detail::check_hip(static_cast<int>(hipEventRecord(ev_main, st)), "event record");
                    detail::check_hip(static_cast<int>(hipStreamWaitEvent(st_side, ev_main, 0)), "stream wait");
                    set_stream(st_side);
                    core.set_wave_priority(true);
                    core.solve_listed(boundary_ids.device_read(), boundary_ids.size(), f, lambda, out);
                    core.set_wave_priority(false);
                    set_stream(st);
                    if (interior_ids.size() > 0)
                        core.solve_listed(interior_ids.device_read(), interior_ids.size(), f, lambda, out);
                    xst = st_side;
                }
                if (world == 1)
                    return;
                const int n_half = n / 2; // = 2 n_shared: lambda half and mu half of a trace vector
                struct Join // the main stream continues after the exchange on the side stream
                {
                    Rank &r;
                    bool on;
                    ~Join()
                    {
                        if (on)
                        {
                            (void)hipEventRecord(r.ev_side, r.st_side);
                            (void)hipStreamWaitEvent(r.st, r.ev_side, 0);
                        }
                    }
                } join{*this, split};
                for (auto &kv : send_slots)
                {
                    const int cnt = kv.second.size();
                    detail::check_hip(cuddh_hip_trace_pack_f32(cnt, n_half, kv.second.device_read(), out, sbuf[kv.first].device_write(), 1, xst),
                                      "trace pack");
                }
                if (loop)
                {
                    detail::check_hip(cuddh_hip_stream_sync(xst), "stream sync"); // my messages are packed
                    for (auto &kv : send_slots)
                        loop->box[static_cast<std::size_t>(rank) * world + kv.first] = sbuf[kv.first].device_read();
                    loop->barrier();
                    for (auto &kv : recv_slots)
                        detail::check_hip(static_cast<int>(hipMemcpyAsync(rbuf[kv.first].device_write(), loop->box[static_cast<std::size_t>(kv.first) * world + rank],
                                                                            sizeof(float) * 2 * kv.second.size(), hipMemcpyDeviceToDevice, xst)),
                                          "loopback message");
                    detail::check_hip(cuddh_hip_stream_sync(xst), "stream sync");
                    loop->barrier(); // every message is delivered: the send buffers may be reused
                    for (auto &kv : recv_slots)
                        detail::check_hip(cuddh_hip_trace_unpack_f32(kv.second.size(), n_half, kv.second.device_read(), rbuf[kv.first].device_read(), out, xst),
                                          "trace unpack");
                    return;
                }
                check_nccl(Rccl::get().GroupStart(), "ncclGroupStart");
                for (int peer = 0; peer < world; ++peer)
                {
                    auto s = send_slots.find(peer);
                    if (s != send_slots.end())
                        check_nccl(Rccl::get().Send(sbuf[peer].device_read(), 2 * static_cast<size_t>(s->second.size()), ncclFloat, peer, comm, xst), "ncclSend");
                    auto r = recv_slots.find(peer);
                    if (r != recv_slots.end())
                        check_nccl(Rccl::get().Recv(rbuf[peer].device_write(), 2 * static_cast<size_t>(r->second.size()), ncclFloat, peer, comm, xst), "ncclRecv");
                }
                check_nccl(Rccl::get().GroupEnd(), "ncclGroupEnd");
                for (auto &kv : recv_slots)
                    detail::check_hip(cuddh_hip_trace_unpack_f32(kv.second.size(), n_half, kv.second.device_read(), rbuf[kv.first].device_read(), out, xst),
                                      "trace unpack");
            }

            void all_reduce(void *d, size_t count, ncclDataType_t type) const { all_reduce_on(loop, comm, st, rank, world, d, count, type); }
        };

        // (I - T) on the partitioned trace vectors of one rank
        class ShardOperator : public SinglePrecisionOperator
        {
        public:
            explicit ShardOperator(Rank &r_) : r(r_) {}
            void action(const float *x, float *y) const override
            {
                r.traces(nullptr, x, y);
                axpby(r.F->size(), 1.0f, x, -1.0f, y);
            }

        private:
            Rank &r;
        };

        void reduce_hook(void *user, void *d_scalars, int count, int is_f64)
        {
            static_cast<Rank *>(user)->all_reduce(d_scalars, static_cast<size_t>(count), is_f64 ? ncclDouble : ncclFloat);
        }
    } // namespace

    multi_gpu_result ddh_solve_multi_gpu(int nx, int nb, double omega, const double *h_a, const double *h_f, double *h_u, int world,
                                         int gmres_m, int gmres_maxit, float tol, int transport, bool split_schedule, int grid_x, int grid_y)
    {
        if ((grid_x != 0 || grid_y != 0) && grid_x * grid_y != world)
            cuddh_error("ddh_solve_multi_gpu error: grid_x * grid_y must be the number of ranks.");
        int n_dev = 0;
        detail::check_hip(static_cast<int>(hipGetDeviceCount(&n_dev)), "hipGetDeviceCount");
        const bool loopback = transport == 2;
        if (world < 1 || (!loopback && world > n_dev) || world > 64)
            cuddh_error("ddh_solve_multi_gpu error: need 1 <= world <= number of visible devices (one rank per GPU).");
        const bool use_rccl = !loopback && (world > 1 || transport == 1);
        Loopback loop_state(world);

        std::vector<ncclComm_t> comms = make_comms(world, use_rccl);

        multi_gpu_result res;
        res.world = world;
        res.used_rccl = use_rccl;
        std::vector<double> t_setup(world, 0), t_rhs(world, 0), t_gmres(world, 0), t_post(world, 0);
        std::vector<solver_out> outs(world);
        std::mutex host_out; // h_u is written by rank 0 only; the mutex guards nothing else
        // test hook (tests/test_gpu_native_drivers.py): CUDDH_MULTIGPU_FAIL_RANK=r makes rank r throw after its right-hand side
        const char *fail_env = std::getenv("CUDDH_MULTIGPU_FAIL_RANK");
        const int fail_rank = fail_env ? std::atoi(fail_env) : -1;

        auto body = [&](int rank)
        {
            {
                detail::check_hip(static_cast<int>(hipSetDevice(loopback ? 0 : rank)), "hipSetDevice");
                Rank R;
                R.rank = rank;
                R.world = world;
                R.device = loopback ? 0 : rank;
                R.comm = comms[rank];
                R.loop = (loopback && world > 1) ? &loop_state : nullptr;
                detail::check_hip(static_cast<int>(hipStreamCreateWithFlags(&R.st, hipStreamNonBlocking)), "hipStreamCreate");
                set_stream(R.st); // thread-local: this thread's launches go to this device's stream

                auto t0 = clk::now();
                R.mesh.reset(new Mesh2D(Mesh2D::uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)));
                R.basis.reset(new Basis(nb));
                R.fem.reset(new H1Space(*R.mesh, *R.basis));
                const int ndof = R.fem->size();
                R.F.reset(new DDH(omega, h_a, *R.fem, nx, nx));
                const auto &core = R.F->internals();
                const int n = R.F->size();
                std::vector<int> dom_rank;
                if (grid_x > 0)
                {
                    const int ndx = nx / (16 / nb); // blocks of 16 / n_basis elements per side (source/DDH.cpp:336)
                    dom_rank = TraceExchangePlan::rank_grid(ndx, core.num_domains() / ndx, grid_x, grid_y);
                }
                R.plan = TraceExchangePlan::build(core.table_B().host_read(), core.num_domains(), core.max_fdof(), n / 2, rank, world,
                                                  dom_rank.empty() ? nullptr : dom_rank.data());
                auto upload = [](const std::vector<int> &v, HostDeviceArray<int> &dst)
                {
                    dst.resize(static_cast<int>(v.size()));
                    std::copy(v.begin(), v.end(), dst.host_write());
                };
                for (const auto &kv : R.plan.send)
                {
                    upload(kv.second, R.send_slots[kv.first]);
                    R.sbuf[kv.first].resize(2 * static_cast<int>(kv.second.size()));
                }
                for (const auto &kv : R.plan.recv)
                {
                    upload(kv.second, R.recv_slots[kv.first]);
                    R.rbuf[kv.first].resize(2 * static_cast<int>(kv.second.size()));
                }
                if (rank == 0)
                    for (const auto &kv : R.plan.send)
                        res.bytes_sent_per_action_rank0 += 2LL * kv.second.size() * sizeof(float);
                const bool one_range = static_cast<int>(R.plan.domains.size()) == R.plan.dom_end - R.plan.dom_begin;
                if (!one_range)
                    upload(R.plan.domains, R.all_ids);
                if (split_schedule && world > 1)
                {
                    detail::check_hip(static_cast<int>(hipStreamCreateWithFlags(&R.st_side, hipStreamNonBlocking)), "hipStreamCreate");
                    detail::check_hip(static_cast<int>(hipEventCreateWithFlags(&R.ev_main, hipEventDisableTiming)), "hipEventCreate");
                    detail::check_hip(static_cast<int>(hipEventCreateWithFlags(&R.ev_side, hipEventDisableTiming)), "hipEventCreate");
                    upload(R.plan.boundary, R.boundary_ids);
                    upload(R.plan.interior, R.interior_ids);
                }

                host_device_dvec f(2 * ndof), u(2 * ndof);
                std::memcpy(f.host_write(), h_f, sizeof(double) * 2 * ndof);
                const double *d_f = f.device_read();
                HostDeviceArray<float> b(n), lam(n);
                float *d_b = b.device_write(), *d_lam = lam.device_write();
                R.sync();
                t_setup[rank] = since(t0);

                t0 = clk::now();
                R.traces(d_f, nullptr, d_b); // DDH::rhs on the partitioned vectors
                R.sync();
                t_rhs[rank] = since(t0);
                if (rank == fail_rank)
                    throw std::runtime_error("multi-GPU solve: injected failure (CUDDH_MULTIGPU_FAIL_RANK)");

                ShardOperator A(R);
                const ScalarReduce red{reduce_hook, &R};
                t0 = clk::now();
                outs[rank] = (use_rccl || R.loop) ? gmres(n, d_lam, &A, d_b, gmres_m, gmres_maxit, tol, 0, 6 * 60 * 60.0, red)
                                      : gmres(n, d_lam, &A, d_b, gmres_m, gmres_maxit, tol, 0);
                R.sync();
                t_gmres[rank] = since(t0);

                t0 = clk::now();
                double *d_u = u.device_write();
                if (R.all_ids.size() > 0) // a rectangle of the block grid: one listed launch
                    R.F->internals().solve_listed(R.all_ids.device_read(), R.all_ids.size(), d_f, d_u, true, d_lam, nullptr);
                else
                    R.F->local_solution(R.plan.dom_begin, R.plan.dom_end, d_lam, d_f, d_u, true);
                R.all_reduce(d_u, 2 * static_cast<size_t>(ndof), ncclDouble); // partition-of-unity sums cross the pieces
                R.sync();
                t_post[rank] = since(t0);
                if (rank == 0)
                {
                    std::lock_guard<std::mutex> lock(host_out);
                    std::memcpy(h_u, u.host_read(), sizeof(double) * 2 * ndof);
                }
            }
        };
        run_ranks(world, loop_state, comms, body);

        res.gmres = outs[0];
        res.t_setup = *std::max_element(t_setup.begin(), t_setup.end());
        res.t_rhs = *std::max_element(t_rhs.begin(), t_rhs.end());
        res.t_gmres = *std::max_element(t_gmres.begin(), t_gmres.end());
        res.t_postprocess = *std::max_element(t_post.begin(), t_post.end());
        return res;
    }
    // ================================================================================== global operator apply over several devices
    namespace
    {
        // One rank of the partitioned fused Helmholtz operator (partition.hpp): its sub-mesh operator plus the two halo exchanges.
        struct HelmRank
        {
            int rank = 0, world = 1;
            Loopback *loop = nullptr;
            ncclComm_t comm = nullptr;
            hipStream_t st = nullptr;
            std::unique_ptr<Mesh2D> gmesh;
            std::unique_ptr<Basis> basis;
            std::unique_ptr<H1Space> gfem;
            std::unique_ptr<FaceSpace> gfs;
            HelmholtzPartition part;
            std::unique_ptr<HelmholtzOperator> op;
            // per exchange: concatenated local-dof lists (peers in increasing rank), the peers' (rank, count) pieces, one buffer each way
            struct Lists
            {
                HostDeviceArray<int> ids;
                std::vector<std::pair<int, int>> pieces; // (peer, number of dofs)
                int total = 0;
            } own, halo;
            host_device_dvec sbuf, rbuf, xs;
            HostDeviceArray<int> halo_all;

            HelmRank() = default;
            HelmRank(const HelmRank &) = delete;
            ~HelmRank()
            {
                op.reset();
                part = HelmholtzPartition();
                gfs.reset();
                gfem.reset();
                if (st)
                {
                    (void)hipStreamSynchronize(st);
                    (void)hipStreamDestroy(st);
                }
            }

            static void concat(const std::map<int, std::vector<int>> &m, Lists &L)
            {
                for (const auto &kv : m)
                {
                    L.pieces.emplace_back(kv.first, static_cast<int>(kv.second.size()));
                    L.total += static_cast<int>(kv.second.size());
                }
                L.ids.resize(std::max(L.total, 1));
                int *h = L.ids.host_write();
                int o = 0;
                for (const auto &kv : m)
                    for (const int l : kv.second)
                        h[o++] = l;
            }

            void sync() const { detail::check_hip(cuddh_hip_stream_sync(st), "stream sync"); }

            // sends `out` pieces of sbuf, receives `in` pieces into rbuf; a piece of k dofs is 2 k doubles (pairs (u, v))
            void exchange(const Lists &out, const Lists &in)
            {
                const double *sb = sbuf.device_read();
                double *rb = rbuf.device_write();
                if (loop)
                {
                    sync(); // my messages are packed
                    int o = 0;
                    for (const auto &pc : out.pieces)
                    {
                        loop->box[static_cast<std::size_t>(rank) * world + pc.first] = sb + 2 * static_cast<std::size_t>(o);
                        o += pc.second;
                    }
                    loop->barrier();
                    o = 0;
                    for (const auto &pc : in.pieces)
                    {
                        detail::check_hip(static_cast<int>(hipMemcpyAsync(rb + 2 * static_cast<std::size_t>(o), loop->box[static_cast<std::size_t>(pc.first) * world + rank],
                                                                            sizeof(double) * 2 * pc.second, hipMemcpyDeviceToDevice, st)),
                                          "loopback message");
                        o += pc.second;
                    }
                    sync();
                    loop->barrier(); // every message is delivered: the send buffer may be reused
                    return;
                }
                check_nccl(Rccl::get().GroupStart(), "ncclGroupStart");
                int so = 0, ro = 0;
                auto s_it = out.pieces.begin();
                auto r_it = in.pieces.begin();
                for (int peer = 0; peer < world; ++peer)
                {
                    if (s_it != out.pieces.end() && s_it->first == peer)
                    {
                        check_nccl(Rccl::get().Send(sb + 2 * static_cast<std::size_t>(so), 2 * static_cast<size_t>(s_it->second), ncclDouble, peer, comm, st), "ncclSend");
                        so += s_it->second;
                        ++s_it;
                    }
                    if (r_it != in.pieces.end() && r_it->first == peer)
                    {
                        check_nccl(Rccl::get().Recv(rb + 2 * static_cast<std::size_t>(ro), 2 * static_cast<size_t>(r_it->second), ncclDouble, peer, comm, st), "ncclRecv");
                        ro += r_it->second;
                        ++r_it;
                    }
                }
                check_nccl(Rccl::get().GroupEnd(), "ncclGroupEnd");
            }

            /// y = A x on this rank's owned entries: halo entries of x are fetched from their owners, those of y end zero
            void apply(const double *x, double *y)
            {
                const int n_loc = part.n_loc;
                double *d_xs = xs.device_write();
                detail::check_hip(cuddh_hip_copy_d2d(d_xs, x, sizeof(double) * 2 * n_loc, st), "halo scratch");
                if (world > 1)
                {
                    detail::check_hip(cuddh_hip_halo_pack_f64(own.total, n_loc, own.ids.device_read(), d_xs, sbuf.device_write(), 0, st), "halo pack x");
                    exchange(own, halo);
                    detail::check_hip(cuddh_hip_halo_unpack_f64(halo.total, n_loc, halo.ids.device_read(), rbuf.device_read(), d_xs, 0, st), "halo unpack x");
                }
                op->action(d_xs, y);
                if (world > 1)
                {
                    detail::check_hip(cuddh_hip_halo_pack_f64(halo.total, n_loc, halo.ids.device_read(), y, sbuf.device_write(), 1, st), "halo pack y");
                    exchange(halo, own);
                    // one launch per sender, in rank order: a dof several ranks hold as halo receives its partial sums in a fixed order
                    int o = 0;
                    for (const auto &pc : own.pieces)
                    {
                        detail::check_hip(cuddh_hip_halo_unpack_f64(pc.second, n_loc, own.ids.device_read() + o, rbuf.device_read() + 2 * static_cast<std::size_t>(o), y, 1, st),
                                          "halo unpack y");
                        o += pc.second;
                    }
                }
            }
        };

        class HelmShardOperator : public Operator
        {
        public:
            explicit HelmShardOperator(HelmRank &r_) : r(r_) {}
            void action(const double *x, double *y) const override { r.apply(x, y); }
            void action(double, const double *, double *) const override { cuddh_error("partitioned Helmholtz operator: action(c, x, y) not implemented"); }

        private:
            HelmRank &r;
        };

        void helm_reduce_hook(void *user, void *d_scalars, int count, int is_f64)
        {
            HelmRank *r = static_cast<HelmRank *>(user);
            all_reduce_on(r->loop, r->comm, r->st, r->rank, r->world, d_scalars, static_cast<size_t>(count), is_f64 ? ncclDouble : ncclFloat);
        }
    } // namespace

    helmholtz_multi_gpu_result helmholtz_multi_gpu(int n_pts, const double *h_xy, int n_elem, const int *h_elems, int nb, double omega, const double *h_a2x,
                                                   const double *h_ax, const double *h_x, double *h_y, int world, int transport, int reps, int gmres_m,
                                                   int gmres_maxit, double tol)
    {
        int n_dev = 0;
        detail::check_hip(static_cast<int>(hipGetDeviceCount(&n_dev)), "hipGetDeviceCount");
        const bool loopback = transport == 2;
        if (world < 1 || (!loopback && world > n_dev) || world > 64)
            cuddh_error("helmholtz_multi_gpu error: need 1 <= world <= number of visible devices (one rank per GPU).");
        const bool use_rccl = !loopback && (world > 1 || transport == 1);
        Loopback loop_state(world);
        std::vector<ncclComm_t> comms = make_comms(world, use_rccl);

        helmholtz_multi_gpu_result res;
        res.world = world;
        res.used_rccl = use_rccl;
        std::vector<double> t_setup(world, 0), t_apply(world, 0), t_gmres(world, 0);
        std::vector<solver_out> outs(world);
        std::vector<long long> n_loc(world, 0), n_halo(world, 0), halo_bytes(world, 0);
        std::vector<std::vector<double>> pieces(world); // every rank's owned entries of the result, assembled by the caller's thread afterwards
        std::vector<std::vector<int>> piece_dofs(world);
        int ndof_global = 0;

        auto body = [&](int rank)
        {
            detail::check_hip(static_cast<int>(hipSetDevice(loopback ? 0 : rank)), "hipSetDevice");
            HelmRank R;
            R.rank = rank;
            R.world = world;
            R.comm = comms[rank];
            R.loop = (loopback && world > 1) ? &loop_state : nullptr;
            detail::check_hip(static_cast<int>(hipStreamCreateWithFlags(&R.st, hipStreamNonBlocking)), "hipStreamCreate");
            set_stream(R.st);

            auto t0 = clk::now();
            R.gmesh.reset(new Mesh2D(Mesh2D::from_vertices(n_pts, h_xy, n_elem, h_elems)));
            R.basis.reset(new Basis(nb));
            R.gfem.reset(new H1Space(*R.gmesh, *R.basis));
            std::vector<int> bfaces;
            for (int e = 0; e < R.gmesh->n_edges(); ++e)
                if (R.gmesh->edge(e)->type == FaceType::BOUNDARY)
                    bfaces.push_back(e);
            R.gfs.reset(new FaceSpace(*R.gfem, static_cast<int>(bfaces.size()), bfaces.data()));
            const int ndof = R.gfem->size();
            if (rank == 0)
                ndof_global = ndof;
            R.part = HelmholtzPartition::build(*R.gmesh, *R.basis, *R.gfem, *R.gfs, rank, world);
            const HelmholtzPartition &P = R.part;
            const int nl = P.n_loc;
            host_device_dvec a2(nl), ax(std::max(P.fs->size(), 1));
            {
                double *h = a2.host_write();
                for (int l = 0; l < nl; ++l)
                    h[l] = h_a2x[P.l2g[l]];
                double *hf = ax.host_write();
                for (int i = 0; i < P.fs->size(); ++i)
                    hf[i] = h_ax[P.face_l2g[i]];
            }
            R.op.reset(new HelmholtzOperator(omega, a2.device_read(), ax.device_read(), *P.fem, *P.fs));
            HelmRank::concat(P.own_to, R.own);
            HelmRank::concat(P.halo_from, R.halo);
            const int mx = std::max(std::max(R.own.total, R.halo.total), 1);
            R.sbuf.resize(2 * mx);
            R.rbuf.resize(2 * mx);
            R.xs.resize(2 * nl);
            n_loc[rank] = nl;
            n_halo[rank] = static_cast<long long>(P.halo.size());
            halo_bytes[rank] = 16LL * (R.own.total + R.halo.total); // sent per apply: x to the holders + partial sums to the owners

            // local vectors: the owned entries of the global input, zero at halo entries
            host_device_dvec x(2 * nl), y(2 * nl);
            {
                double *h = x.host_write();
                std::fill(h, h + 2 * nl, 0.0);
                for (const int l : P.owned)
                {
                    h[l] = h_x[P.l2g[l]];
                    h[nl + l] = h_x[ndof + P.l2g[l]];
                }
            }
            const double *d_x = x.device_read();
            double *d_y = y.device_write();
            R.sync();
            t_setup[rank] = since(t0);

            if (gmres_maxit > 0)
            {
                HelmShardOperator A(R);
                const ScalarReduce red{helm_reduce_hook, &R};
                zeros(2 * nl, d_y);
                t0 = clk::now();
                outs[rank] = (use_rccl || R.loop) ? gmres(2 * nl, d_y, &A, d_x, gmres_m, gmres_maxit, tol, 0, 6 * 60 * 60.0, red)
                                                  : gmres(2 * nl, d_y, &A, d_x, gmres_m, gmres_maxit, tol, 0);
                R.sync();
                t_gmres[rank] = since(t0);
            }
            else
            {
                R.apply(d_x, d_y);
                R.sync();
                if (reps > 0)
                {
                    if (R.loop)
                        R.loop->barrier();
                    t0 = clk::now();
                    for (int i = 0; i < reps; ++i)
                        R.apply(d_x, d_y);
                    R.sync();
                    t_apply[rank] = since(t0) / reps;
                }
            }
            const double *hy = y.host_read();
            pieces[rank].resize(2 * P.owned.size());
            piece_dofs[rank].resize(P.owned.size());
            for (std::size_t i = 0; i < P.owned.size(); ++i)
            {
                piece_dofs[rank][i] = P.l2g[P.owned[i]];
                pieces[rank][2 * i] = hy[P.owned[i]];
                pieces[rank][2 * i + 1] = hy[nl + P.owned[i]];
            }
        };
        run_ranks(world, loop_state, comms, body);

        // every dof has exactly one owner: the result is the union of the ranks' owned entries
        for (int r = 0; r < world; ++r)
            for (std::size_t i = 0; i < piece_dofs[r].size(); ++i)
            {
                h_y[piece_dofs[r][i]] = pieces[r][2 * i];
                h_y[ndof_global + piece_dofs[r][i]] = pieces[r][2 * i + 1];
            }
        res.gmres = outs[0];
        res.t_setup = *std::max_element(t_setup.begin(), t_setup.end());
        res.t_apply = *std::max_element(t_apply.begin(), t_apply.end());
        res.t_gmres = *std::max_element(t_gmres.begin(), t_gmres.end());
        res.n_loc_max = *std::max_element(n_loc.begin(), n_loc.end());
        res.n_halo_max = *std::max_element(n_halo.begin(), n_halo.end());
        res.halo_bytes_per_apply_max = *std::max_element(halo_bytes.begin(), halo_bytes.end());
        return res;
    }
} // namespace cuddh
