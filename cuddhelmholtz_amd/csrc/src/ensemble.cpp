// Construction of the subdomain gather/scatter maps.
#include "cuddh/ensemble.hpp"

#include <algorithm>
#include <array>
#include <cstdint>
#include <unordered_set>
#include <utility>
#include <vector>

namespace cuddh
{
    namespace
    {
        struct SubFace
        {
            int edge; // global edge id
            int side; // 0: the subspace holds elements[0] of the edge, 1: elements[1]
        };

        template <typename T>
        int longest(const std::vector<std::vector<T>> &lists)
        {
            std::size_t m = 0;
            for (const auto &l : lists)
                m = std::max(m, l.size());
            return static_cast<int>(m);
        }
    } // namespace

    EnsembleSpace::EnsembleSpace(const H1Space &fem, int n_spaces_, const int *labels)
        : n_spaces(n_spaces_), n_basis(fem.basis().size()), s_dof(n_spaces_), s_elems(n_spaces_), s_faces(n_spaces_),
          s_fdof(n_spaces_)
    {
        const Mesh2D &mesh = fem.mesh();
        const int nel = mesh.n_elem();
        const int nb = n_basis;
        const int g_ndof = fem.size();

        // ---- elements of each subspace, in increasing global order
        std::vector<std::vector<int>> members(n_spaces);
        std::vector<int> local_elem(nel);
        for (int el = 0; el < nel; ++el)
        {
            const int p = labels[el];
            if (p < 0 || p >= n_spaces)
                cuddh_error("EnsembleSpace error: an element was illogically labeled.");
            local_elem[el] = static_cast<int>(members[p].size());
            members[p].push_back(el);
        }
        for (const auto &mlist : members)
            if (mlist.empty())
                cuddh_error("EnsembleSpace error: atleast one space is empty");
        mx_elems = longest(members);

        int *h_s_elems = s_elems.host_write();
        elems.resize(mx_elems * n_spaces);
        int *h_elems = elems.host_write();
        std::fill(h_elems, h_elems + mx_elems * n_spaces, -1);
        for (int p = 0; p < n_spaces; ++p)
        {
            h_s_elems[p] = static_cast<int>(members[p].size());
            std::copy(members[p].begin(), members[p].end(), h_elems + mx_elems * p);
        }

        // ---- boundary faces of each subspace (global edge order) and the faces two subspaces share
        std::vector<std::vector<SubFace>> sub_faces(n_spaces);
        std::vector<std::array<int, 4>> shared_faces; // {S0, S1, face index in S0, face index in S1}
        const int g_edges = mesh.n_edges();
        for (int e = 0; e < g_edges; ++e)
        {
            const Edge *edge = mesh.edge(e);
            const int S0 = labels[edge->elements[0]];
            if (edge->type == FaceType::BOUNDARY)
            {
                sub_faces[S0].push_back({e, 0});
                continue;
            }
            const int S1 = labels[edge->elements[1]];
            if (S0 == S1)
                continue;
            sub_faces[S0].push_back({e, 0});
            sub_faces[S1].push_back({e, 1});
            shared_faces.push_back({S0, S1, static_cast<int>(sub_faces[S0].size()) - 1, static_cast<int>(sub_faces[S1].size()) - 1});
        }
        mx_faces = longest(sub_faces);

        int *h_s_faces = s_faces.host_write();
        _faces.resize(mx_faces * n_spaces);
        int *h_faces = _faces.host_write();
        std::fill(h_faces, h_faces + mx_faces * n_spaces, -1);
        for (int p = 0; p < n_spaces; ++p)
        {
            h_s_faces[p] = static_cast<int>(sub_faces[p].size());
            for (std::size_t f = 0; f < sub_faces[p].size(); ++f)
                h_faces[f + static_cast<std::size_t>(mx_faces) * p] = sub_faces[p][f].edge;
        }

        // ---- subspace dof numbering: first touch over (el, j, i)
        sI.resize(nb * nb * mx_elems * n_spaces);
        int *h_sI = sI.host_write();
        std::fill(h_sI, h_sI + static_cast<std::size_t>(nb) * nb * mx_elems * n_spaces, -1);
        const int *g_inds = fem.global_indices(MemorySpace::HOST);

        std::vector<std::vector<int>> sub_to_global(n_spaces);
        // stamp[g] == p  <=>  global dof g already has local number slot[g] in subspace p
        std::vector<int> stamp(g_ndof, -1), slot(g_ndof, -1);
        int *h_s_dof = s_dof.host_write();
        for (int p = 0; p < n_spaces; ++p)
        {
            auto &s2g = sub_to_global[p];
            for (int el = 0; el < h_s_elems[p]; ++el)
            {
                const int *gi = g_inds + static_cast<std::size_t>(nb) * nb * members[p][el];
                int *si = h_sI + static_cast<std::size_t>(nb) * nb * (el + static_cast<std::size_t>(mx_elems) * p);
                for (int v = 0; v < nb * nb; ++v)
                {
                    const int g = gi[v];
                    if (stamp[g] != p)
                    {
                        stamp[g] = p;
                        slot[g] = static_cast<int>(s2g.size());
                        s2g.push_back(g);
                    }
                    si[v] = slot[g];
                }
            }
            h_s_dof[p] = static_cast<int>(s2g.size());
        }
        mx_ndof = longest(sub_to_global);

        gI.resize(mx_ndof * n_spaces);
        int *h_gI = gI.host_write();
        std::fill(h_gI, h_gI + static_cast<std::size_t>(mx_ndof) * n_spaces, -1);
        for (int p = 0; p < n_spaces; ++p)
            std::copy(sub_to_global[p].begin(), sub_to_global[p].end(), h_gI + static_cast<std::size_t>(mx_ndof) * p);

        // ---- face-space numbering: first touch over (face, i)
        fI.resize(nb * mx_faces * n_spaces);
        int *h_fI = fI.host_write();
        std::fill(h_fI, h_fI + static_cast<std::size_t>(nb) * mx_faces * n_spaces, -1);

        std::vector<std::vector<int>> face_to_sub(n_spaces);
        std::vector<int> fslot(mx_ndof);
        int *h_s_fdof = s_fdof.host_write();
        for (int p = 0; p < n_spaces; ++p)
        {
            auto &f2s = face_to_sub[p];
            std::fill(fslot.begin(), fslot.end(), -1);
            for (int f = 0; f < h_s_faces[p]; ++f)
            {
                const SubFace sf = sub_faces[p][f];
                const Edge *edge = mesh.edge(sf.edge);
                const int g_el = edge->elements[sf.side];
                const int s = edge->sides[sf.side];
                const bool flip = (sf.side == 1 && edge->delta < 0);
                const int *si = h_sI + static_cast<std::size_t>(nb) * nb * (local_elem[g_el] + static_cast<std::size_t>(mx_elems) * p);
                for (int i = 0; i < nb; ++i)
                {
                    const int t = flip ? nb - 1 - i : i;
                    int a, b;
                    switch (s)
                    {
                    case 0: a = t; b = 0; break;
                    case 1: a = nb - 1; b = t; break;
                    case 2: a = t; b = nb - 1; break;
                    default: a = 0; b = t; break;
                    }
                    const int d = si[a + nb * b];
                    if (fslot[d] < 0)
                    {
                        fslot[d] = static_cast<int>(f2s.size());
                        f2s.push_back(d);
                    }
                    h_fI[i + nb * (f + static_cast<std::size_t>(mx_faces) * p)] = fslot[d];
                }
            }
            h_s_fdof[p] = static_cast<int>(f2s.size());
        }
        mx_fdof = longest(face_to_sub);

        pI.resize(mx_fdof * n_spaces);
        int *h_pI = pI.host_write();
        std::fill(h_pI, h_pI + static_cast<std::size_t>(mx_fdof) * n_spaces, -1);
        for (int p = 0; p < n_spaces; ++p)
            std::copy(face_to_sub[p].begin(), face_to_sub[p].end(), h_pI + static_cast<std::size_t>(mx_fdof) * p);

        // ---- shared dof pairs, unique per (unordered subspace pair, dof of the lower-numbered subspace)
        std::vector<std::array<int, 4>> pairs;
        std::unordered_set<std::uint64_t> seen;
        seen.reserve(shared_faces.size() * nb);
        for (const auto &sfc : shared_faces)
        {
            const int S0 = sfc[0], S1 = sfc[1], f0 = sfc[2], f1 = sfc[3];
            const std::uint64_t lo = std::min(S0, S1), hi = std::max(S0, S1);
            const std::uint64_t pair_id = lo + static_cast<std::uint64_t>(n_spaces) * hi;
            for (int i = 0; i < nb; ++i)
            {
                const int j0 = h_fI[i + nb * (f0 + static_cast<std::size_t>(mx_faces) * S0)];
                const int j1 = h_fI[i + nb * (f1 + static_cast<std::size_t>(mx_faces) * S1)];
                const std::uint64_t dof_of_lower = static_cast<std::uint64_t>(S0 < S1 ? j0 : j1);
                if (seen.insert(pair_id * static_cast<std::uint64_t>(mx_fdof + 1) + dof_of_lower).second)
                    pairs.push_back({S0, S1, j0, j1});
            }
        }

        n_shared_dofs = static_cast<int>(pairs.size());
        cmap.resize(4 * n_shared_dofs);
        int *h_cmap = cmap.host_write();
        for (int k = 0; k < n_shared_dofs; ++k)
            for (int c = 0; c < 4; ++c)
                h_cmap[c + 4 * k] = pairs[k][c];
    }
} // namespace cuddh
