// DDH over the GPUs of one node from ONE process: the subdomains are cut into `world` pieces -- contiguous ranges
// (strips of block rows) or the rectangles of a gx x gy rank grid --, one per device; every device runs its piece of the local solves on its own stream, the traces its subdomains write for
// subdomains of another piece travel by one grouped RCCL send/recv per action (ncclGroupStart / ncclSend / ncclRecv /
// ncclGroupEnd over xGMI -- the neighbour all-to-all of the north star), and GMRES runs on partitioned trace vectors with
// every inner product summed by ncclAllReduce through the ScalarReduce hook of krylov.hpp, so all devices take identical
// decisions.  The reference is single-GPU; this layer is new.  The Python host (cuddhelmholtz_amd/dist.py, one process per
// GPU over torch.distributed) implements the same scheme; TraceExchangePlan here and TraceExchange there produce identical
// ownership / send / receive lists (tests/test_distributed_gloo.py).
#ifndef CUDDH_AMD_MULTIGPU_HPP
#define CUDDH_AMD_MULTIGPU_HPP

#include <map>
#include <memory>
#include <vector>

#include "krylov.hpp"

namespace cuddh
{
    /// contiguous balanced range of n_items for `rank` of `world`
    inline void shard_range(int n_items, int rank, int world, int &begin, int &end)
    {
        begin = static_cast<int>((static_cast<long long>(n_items) * rank) / world);
        end = static_cast<int>((static_cast<long long>(n_items) * (rank + 1)) / world);
    }

    /// Who owns, sends and receives which trace slots, from the slot table B (mx_fdof, 2, n_domains) of the DDH constructor
    /// (reference source/DDH.cpp:425-440: B(i,0,S) is the slot subdomain S reads for its face dof i, B(i,1,S) the slot it
    /// writes).  owner(slot) = rank of the subdomain that reads it, else of the one that writes it; slots nobody touches
    /// (the reference's orphan slots at cross points) belong to nobody and stay zero.  Host-only.
    struct TraceExchangePlan
    {
        int rank = 0, world = 1, n_lambda = 0, dom_begin = 0, dom_end = 0;
        std::vector<int> owned;                // slots this rank owns, increasing
        std::map<int, std::vector<int>> send;  // peer -> slots this rank's subdomains write and the peer owns, increasing
        std::map<int, std::vector<int>> recv;  // peer -> slots this rank owns and the peer's subdomains write, increasing
        /// split schedule: the subdomains whose traces other ranks wait for (plus up to 7 others, so that both launches consist
        /// of whole workgroups of the wavefront kernels) and the rest of [dom_begin, dom_end); both increasing
        std::vector<int> boundary, interior;
        /// this rank's subdomains, increasing ([dom_begin, dom_end) for strips; a rectangle's rows for a rank grid)
        std::vector<int> domains;

        /// dom_rank (n_domains, or null): subdomain -> rank; null = `world` contiguous ranges (strips of block rows)
        static TraceExchangePlan build(const int *B, int n_domains, int mx_fdof, int n_lambda, int rank, int world,
                                       const int *dom_rank = nullptr);
        /// subdomain -> rank for a gx x gy grid of ranks over the ndx x ndy block grid (rank = rx + gx ry)
        static std::vector<int> rank_grid(int ndx, int ndy, int gx, int gy);
    };

    struct multi_gpu_result
    {
        solver_out gmres;
        double t_setup = 0, t_rhs = 0, t_gmres = 0, t_postprocess = 0; // seconds, max over ranks
        int world = 1;
        bool used_rccl = false;
        long long bytes_sent_per_action_rank0 = 0;
    };

    /// The flow of examples/DDH.cpp:141-144 (rhs -> gmres -> postprocess) on `world` devices of this process: uniform_rect
    /// (nx x nx on [-1,1]^2), Basis(nb), h_a nodal coefficient (HOST, global numbering), h_f = [f; g] load vector (HOST,
    /// 2 ndof), h_u receives [u; v] (HOST, 2 ndof).  transport: 0 = RCCL for world > 1, no communicator for world = 1;
    /// 1 = RCCL also for world = 1 (the one-rank communicator carries the reductions: exercises the RCCL calls on a one-GPU
    /// box); 2 = loopback: the `world` ranks are host threads SHARING device 0, messages are device-to-device copies and
    /// reductions host sums in rank order -- a test transport that runs the whole N > 1 path except the RCCL calls on one GPU.
    /// split_schedule: the boundary subdomains are solved first, as one listed launch with issue priority on a second stream,
    /// the exchange is posted behind them and the interior subdomains run on the main stream meanwhile (the north star's
    /// schedule; default: exchange after all local solves).  grid_x x grid_y = world: the ranks own rectangles of the
    /// subdomain grid instead of strips of block rows (0: strips).
    multi_gpu_result ddh_solve_multi_gpu(int nx, int nb, double omega, const double *h_a, const double *h_f, double *h_u, int world,
                                         int gmres_m, int gmres_maxit, float tol, int transport = 0, bool split_schedule = false,
                                         int grid_x = 0, int grid_y = 0);

    struct helmholtz_multi_gpu_result
    {
        solver_out gmres;
        double t_setup = 0, t_apply = 0, t_gmres = 0; // seconds, max over ranks (t_apply: per apply)
        int world = 1;
        bool used_rccl = false;
        long long n_loc_max = 0, n_halo_max = 0, halo_bytes_per_apply_max = 0;
    };

    /// The fused complex Helmholtz operator of examples/Helmholtz.hpp:28-56 partitioned over `world` devices of this process
    /// (partition.hpp: element partition, sub-mesh operators, two halo exchanges per apply through cuddh_hip_halo_pack / unpack
    /// and grouped RCCL send / recv).  Mesh: n_pts vertices h_xy (2, n_pts), n_elem quadrilaterals h_elems (4, n_elem) as
    /// Mesh2D::from_vertices takes them; Basis(nb); h_a2x (ndof) nodal a^2, h_ax (FaceSpace of all boundary edges) a -- HOST,
    /// global numbering.  gmres_maxit == 0: h_y = A h_x ([u; v], 2 ndof each; the apply is then repeated `reps` times for
    /// t_apply); gmres_maxit > 0: h_y = GMRES(gmres_m) solution of A y = h_x from y = 0, every inner product all-reduced.
    /// transport as in ddh_solve_multi_gpu (0 RCCL, 1 RCCL also for one rank, 2 loopback ranks sharing device 0).
    helmholtz_multi_gpu_result helmholtz_multi_gpu(int n_pts, const double *h_xy, int n_elem, const int *h_elems, int nb, double omega,
                                                   const double *h_a2x, const double *h_ax, const double *h_x, double *h_y, int world,
                                                   int transport = 0, int reps = 0, int gmres_m = 20, int gmres_maxit = 0, double tol = 1e-8);
} // namespace cuddh

#endif
