// Global numbering of the H1 space and of face spaces.
#include "cuddh/spaces.hpp"

#include <atomic>
#include <vector>

#include "cuddh/parallel.hpp"
#include "cuddh_hip.h"

namespace cuddh
{
    namespace
    {
        // flat element-local index of the i-th node (in the side's own direction) on local side s
        inline int side_node(int nb, int i, int s, int el)
        {
            int a, b; // (xi index, eta index)
            switch (s)
            {
            case 0: a = i; b = 0; break;      // eta = -1
            case 1: a = nb - 1; b = i; break; // xi = +1
            case 2: a = i; b = nb - 1; break; // eta = +1
            default: a = 0; b = i; break;     // xi = -1
            }
            return a + nb * (b + nb * el);
        }

        // flat element-local index of corner c
        inline int corner_node(int nb, int c, int el)
        {
            const int a = (c == 0 || c == 3) ? 0 : nb - 1;
            const int b = (c == 0 || c == 1) ? 0 : nb - 1;
            return a + nb * (b + nb * el);
        }
    } // namespace

    H1Space::H1Space(const Mesh2D &mesh_, const Basis &basis_)
        : n_elem(mesh_.n_elem()), n_basis(basis_.size()), _mesh(mesh_), _basis(basis_), ndof(0),
          _I(basis_.size() * basis_.size() * mesh_.n_elem())
    {
        const int nb = n_basis;
        const int N = nb * nb * n_elem;
        int *I = _I.host_write();

        detail::PhaseTimer timer;
        // owner[v] = flat index of the node v is a copy of (-1: v owns itself).  The loops below are the reference's
        // (source/H1Space.cpp:47-106) cut into contiguous ranges for the set-up threads; every entry has one writer.
        std::vector<int> owner(N, -1);

        if (nb > 2)
        {
            const int n_int = _mesh.n_edges(FaceType::INTERIOR);
            detail::parallel_for(static_cast<std::size_t>(n_int), [&](std::size_t e0, std::size_t e1, int)
            {
                for (int e = static_cast<int>(e0); e < static_cast<int>(e1); ++e)
                {
                    const Edge *edge = _mesh.edge(e, FaceType::INTERIOR);
                    const bool flip = edge->delta < 0;
                    for (int i = 1; i < nb - 1; ++i)
                    {
                        const int mine = side_node(nb, i, edge->sides[0], edge->elements[0]);
                        const int theirs = side_node(nb, flip ? nb - 1 - i : i, edge->sides[1], edge->elements[1]);
                        owner[theirs] = mine;
                    }
                }
            });
        }

        const int n_nodes = _mesh.n_nodes();
        detail::parallel_for(static_cast<std::size_t>(n_nodes), [&](std::size_t k0, std::size_t k1, int)
        {
            for (int k = static_cast<int>(k0); k < static_cast<int>(k1); ++k)
            {
                const auto &adj = _mesh.node(k).connected_elements;
                if (adj.empty())
                    continue;
                const int first = corner_node(nb, adj[0].i, adj[0].id);
                for (std::size_t t = 1; t < adj.size(); ++t)
                    owner[corner_node(nb, adj[t].i, adj[t].id)] = first;
            }
        });

        timer.lap("H1Space: owners");
        // owners are numbered in increasing flat index: count per range, prefix sum, assign; then the copies
        const int C = detail::chunk_count(static_cast<std::size_t>(N));
        std::vector<int> first_id(C + 1, 0);
        detail::parallel_for(static_cast<std::size_t>(N), [&](std::size_t v0, std::size_t v1, int c)
        {
            int cnt = 0;
            for (std::size_t v = v0; v < v1; ++v)
                cnt += owner[v] < 0;
            first_id[c + 1] = cnt;
        });
        for (int c = 0; c < C; ++c)
            first_id[c + 1] += first_id[c];
        ndof = first_id[C];
        detail::parallel_for(static_cast<std::size_t>(N), [&](std::size_t v0, std::size_t v1, int c)
        {
            int next = first_id[c];
            for (std::size_t v = v0; v < v1; ++v)
                if (owner[v] < 0)
                    I[v] = next++;
        });
        detail::parallel_for(static_cast<std::size_t>(N), [&](std::size_t v0, std::size_t v1, int)
        {
            for (std::size_t v = v0; v < v1; ++v)
                if (owner[v] >= 0)
                    I[v] = I[owner[v]];
        });

        timer.lap("H1Space: numbering");
    }

    void H1Space::build_collocation_points() const
    {
        detail::PhaseTimer timer;
        const int nb = n_basis;
        const int *I = _I.host_read();
        // collocation points: the reference overwrites a shared dof's point element after element, the highest element wins
        // (source/H1Space.cpp:108-126).  In parallel: each range of elements first stamps the dofs it touches with its range
        // number (highest wins), then writes only the dofs it won, in element order -- the same winner as the serial loop.
        _xy.resize(2 * ndof);
        double *xy = _xy.host_write();
        const QuadratureRule &gll = _basis.quadrature();
        std::vector<std::atomic<int>> winner(ndof);
        detail::parallel_for(static_cast<std::size_t>(ndof), [&](std::size_t g0, std::size_t g1, int)
        {
            for (std::size_t g = g0; g < g1; ++g)
                winner[g].store(-1, std::memory_order_relaxed);
        });
        detail::parallel_for(static_cast<std::size_t>(n_elem), [&](std::size_t e0, std::size_t e1, int c)
        {
            for (std::size_t v = e0 * nb * nb; v < e1 * nb * nb; ++v)
            {
                std::atomic<int> &w = winner[I[v]];
                int cur = w.load(std::memory_order_relaxed);
                while (cur < c && !w.compare_exchange_weak(cur, c, std::memory_order_relaxed))
                {
                }
            }
        }, 64);
        detail::parallel_for(static_cast<std::size_t>(n_elem), [&](std::size_t e0, std::size_t e1, int c)
        {
            for (int el = static_cast<int>(e0); el < static_cast<int>(e1); ++el)
            {
                const Element *elem = _mesh.element(el);
                for (int j = 0; j < nb; ++j)
                    for (int i = 0; i < nb; ++i)
                    {
                        const int g = I[i + nb * (j + nb * el)];
                        if (winner[g].load(std::memory_order_relaxed) != c)
                            continue;
                        const double xi[2] = {gll.x(i), gll.x(j)};
                        elem->physical_coordinates(xi, xy + 2 * g);
                    }
            }
        }, 64);
        timer.lap("H1Space: collocation points");
    }

    FaceSpace::FaceSpace(const H1Space &fem_, int nf, const int *faces_)
        : fem(fem_), _n_faces(nf), n_basis(fem_.basis().size()), ndof(0), _I(fem_.basis().size() * nf), _faces(nf)
    {
        const int nb = n_basis;
        int *F = _faces.host_write();
        int *I = _I.host_write();
        for (int f = 0; f < nf; ++f)
            F[f] = faces_[f];

        const int *K = fem.global_indices(MemorySpace::HOST);
        const Mesh2D &mesh = fem.mesh();

        // first touch over (face, i) numbers the face dofs
        std::vector<int> local_of(fem.size(), -1);
        std::vector<int> proj;
        for (int f = 0; f < nf; ++f)
        {
            const Edge *edge = mesh.edge(F[f]);
            for (int i = 0; i < nb; ++i)
            {
                const int g = K[side_node(nb, i, edge->sides[0], edge->elements[0])];
                if (local_of[g] < 0)
                {
                    local_of[g] = static_cast<int>(proj.size());
                    proj.push_back(g);
                }
                I[i + nb * f] = local_of[g];
            }
        }

        ndof = static_cast<int>(proj.size());
        _proj.resize(ndof);
        int *p = _proj.host_write();
        for (int i = 0; i < ndof; ++i)
            p[i] = proj[i];
    }

    void FaceSpace::restrict(const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_gather_f64(ndof, _proj.device_read(), x, y, stream()), "FaceSpace::restrict");
    }

    void FaceSpace::prolong(const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_scatter_add_f64(ndof, _proj.device_read(), x, y, stream()), "FaceSpace::prolong");
    }

    void FaceSpace::orth(double *x) const
    {
        detail::check_hip(cuddh_hip_zero_indexed_f64(ndof, _proj.device_read(), x, stream()), "FaceSpace::orth");
    }

    const Mesh2D::EdgeMetricCollection &FaceSpace::metrics(const QuadratureRule &quad) const
    {
        auto &slot = _metrics[quad.name()];
        if (!slot)
            slot = std::make_unique<Mesh2D::EdgeMetricCollection>(fem.mesh(), _n_faces, _faces.host_read(), quad);
        return *slot;
    }
} // namespace cuddh
