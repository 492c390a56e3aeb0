// Construction of the subdomain gather/scatter maps.
//
// The rules are the reference's (source/EnsembleSpace.cpp:11-287, see ensemble.hpp); the schedule is not: the reference
// walks everything with per-dof hash maps on one thread.  Here the per-subspace lists are counting-sorted into flat arrays,
// the two numbering passes (subspace dofs, face dofs) run over the subspaces in parallel with a small per-thread lookup
// table, and the shared-dof de-duplication uses one flat open-addressing set.  65,536 subspaces (1024^2 elements): 0.58 s ->
// see profiles/r02/setup_time_1024.txt.  Results do not depend on the thread count.
#include "cuddh/ensemble.hpp"

#include <algorithm>
#include <array>
#include <cstdint>
#include <utility>
#include <vector>

#include "cuddh/parallel.hpp"

namespace cuddh
{
    namespace
    {
        struct SubFace
        {
            int edge; // global edge id
            int side; // 0: the subspace holds elements[0] of the edge, 1: elements[1]
        };

        // first-touch numbering of the keys of ONE subspace: key -> 0, 1, 2, ... in order of first appearance.
        // Open addressing, versioned so that clearing between subspaces is free.
        class FirstTouch
        {
        public:
            explicit FirstTouch(int max_keys)
            {
                cap = 64;
                while (cap < 4 * static_cast<std::size_t>(std::max(1, max_keys)))
                    cap <<= 1;
                key.assign(cap, 0);
                val.assign(cap, 0);
                ver.assign(cap, 0u);
            }

            void reset()
            {
                ++version;
                count = 0;
            }

            /// number of `k` within the current subspace; `fresh` tells whether this call assigned it
            int number(int k, bool &fresh)
            {
                std::size_t h = (static_cast<std::uint32_t>(k) * 2654435761u) & (cap - 1);
                while (ver[h] == version)
                {
                    if (key[h] == k)
                    {
                        fresh = false;
                        return val[h];
                    }
                    h = (h + 1) & (cap - 1);
                }
                ver[h] = version;
                key[h] = k;
                val[h] = count;
                fresh = true;
                return count++;
            }

            int size() const { return count; }

        private:
            std::size_t cap;
            std::vector<int> key, val;
            std::vector<std::uint32_t> ver;
            std::uint32_t version = 0;
            int count = 0;
        };

        // flat set of 64-bit keys (insert only)
        class KeySet
        {
        public:
            explicit KeySet(std::size_t expected)
            {
                cap = 1024;
                while (cap < 2 * expected + 16)
                    cap <<= 1;
                slot.assign(cap, EMPTY);
            }

            /// true when the key was not present
            bool insert(std::uint64_t k)
            {
                if (2 * (used + 1) > cap)
                    grow();
                std::size_t h = static_cast<std::size_t>((k * 0x9E3779B97F4A7C15ull) >> 17) & (cap - 1);
                while (slot[h] != EMPTY)
                {
                    if (slot[h] == k)
                        return false;
                    h = (h + 1) & (cap - 1);
                }
                slot[h] = k;
                ++used;
                return true;
            }

        private:
            void grow()
            {
                std::vector<std::uint64_t> old;
                old.swap(slot);
                cap <<= 1;
                slot.assign(cap, EMPTY);
                used = 0;
                for (const std::uint64_t k : old)
                    if (k != EMPTY)
                        insert(k);
            }

            std::size_t used = 0;
            static constexpr std::uint64_t EMPTY = ~0ull;
            std::size_t cap;
            std::vector<std::uint64_t> slot;
        };
    } // namespace

    EnsembleSpace::EnsembleSpace(const H1Space &fem, int n_spaces_, const int *labels)
        : n_spaces(n_spaces_), n_basis(fem.basis().size()), s_dof(n_spaces_), s_elems(n_spaces_), s_faces(n_spaces_),
          s_fdof(n_spaces_)
    {
        const Mesh2D &mesh = fem.mesh();
        const int nel = mesh.n_elem();
        const int nb = n_basis, nn = nb * nb;
        detail::PhaseTimer timer;

        // ---- elements of each subspace, in increasing global order (counting sort)
        int *h_s_elems = s_elems.host_write();
        std::fill(h_s_elems, h_s_elems + n_spaces, 0);
        std::vector<int> local_elem(nel);
        for (int el = 0; el < nel; ++el)
        {
            const int p = labels[el];
            if (p < 0 || p >= n_spaces)
                cuddh_error("EnsembleSpace error: an element was illogically labeled.");
            local_elem[el] = h_s_elems[p]++;
        }
        mx_elems = 0;
        for (int p = 0; p < n_spaces; ++p)
        {
            if (h_s_elems[p] == 0)
                cuddh_error("EnsembleSpace error: atleast one space is empty");
            mx_elems = std::max(mx_elems, h_s_elems[p]);
        }
        elems.resize(mx_elems * n_spaces);
        int *h_elems = elems.host_write();
        std::fill(h_elems, h_elems + static_cast<std::size_t>(mx_elems) * n_spaces, -1);
        for (int el = 0; el < nel; ++el)
            h_elems[local_elem[el] + static_cast<std::size_t>(mx_elems) * labels[el]] = el;

        timer.lap("EnsembleSpace: elements");
        // ---- boundary faces of each subspace (global edge order) and the faces two subspaces share
        const int g_edges = mesh.n_edges();
        int *h_s_faces = s_faces.host_write();
        std::fill(h_s_faces, h_s_faces + n_spaces, 0);
        for (int e = 0; e < g_edges; ++e)
        {
            const Edge *edge = mesh.edge(e);
            const int S0 = labels[edge->elements[0]];
            if (edge->type == FaceType::BOUNDARY)
            {
                ++h_s_faces[S0];
                continue;
            }
            const int S1 = labels[edge->elements[1]];
            if (S0 != S1)
            {
                ++h_s_faces[S0];
                ++h_s_faces[S1];
            }
        }
        mx_faces = 0;
        std::vector<std::size_t> face_off(static_cast<std::size_t>(n_spaces) + 1, 0);
        for (int p = 0; p < n_spaces; ++p)
        {
            mx_faces = std::max(mx_faces, h_s_faces[p]);
            face_off[p + 1] = face_off[p] + h_s_faces[p];
        }
        std::vector<SubFace> sub_faces(face_off[n_spaces]);
        std::vector<int> fill(n_spaces, 0);
        std::vector<std::array<int, 4>> shared_faces; // {S0, S1, face index in S0, face index in S1}
        for (int e = 0; e < g_edges; ++e)
        {
            const Edge *edge = mesh.edge(e);
            const int S0 = labels[edge->elements[0]];
            if (edge->type == FaceType::BOUNDARY)
            {
                sub_faces[face_off[S0] + fill[S0]++] = {e, 0};
                continue;
            }
            const int S1 = labels[edge->elements[1]];
            if (S0 == S1)
                continue;
            sub_faces[face_off[S0] + fill[S0]] = {e, 0};
            sub_faces[face_off[S1] + fill[S1]] = {e, 1};
            shared_faces.push_back({S0, S1, fill[S0], fill[S1]});
            ++fill[S0];
            ++fill[S1];
        }

        _faces.resize(mx_faces * n_spaces);
        int *h_faces = _faces.host_write();
        std::fill(h_faces, h_faces + static_cast<std::size_t>(mx_faces) * n_spaces, -1);
        for (int p = 0; p < n_spaces; ++p)
            for (int f = 0; f < h_s_faces[p]; ++f)
                h_faces[f + static_cast<std::size_t>(mx_faces) * p] = sub_faces[face_off[p] + f].edge;

        timer.lap("EnsembleSpace: faces");
        // ---- subspace dof numbering: first touch over (el, j, i); subspaces are independent of each other
        sI.resize(nn * mx_elems * n_spaces);
        int *h_sI = sI.host_write();
        const int *g_inds = fem.global_indices(MemorySpace::HOST);
        const std::size_t wide = static_cast<std::size_t>(nn) * mx_elems; // upper bound of a subspace's dofs
        std::vector<int> s2g(wide * n_spaces);                           // subspace dof -> global dof, stride `wide`
        int *h_s_dof = s_dof.host_write();
        detail::parallel_for(static_cast<std::size_t>(n_spaces), [&](std::size_t p0, std::size_t p1, int)
        {
            FirstTouch table(static_cast<int>(wide));
            for (std::size_t p = p0; p < p1; ++p)
            {
                table.reset();
                int *mine = s2g.data() + wide * p;
                int *si = h_sI + wide * p;
                for (int el = 0; el < h_s_elems[p]; ++el)
                {
                    const int *gi = g_inds + static_cast<std::size_t>(nn) * h_elems[el + static_cast<std::size_t>(mx_elems) * p];
                    for (int v = 0; v < nn; ++v)
                    {
                        bool fresh;
                        const int l = table.number(gi[v], fresh);
                        if (fresh)
                            mine[l] = gi[v];
                        si[static_cast<std::size_t>(nn) * el + v] = l;
                    }
                }
                for (std::size_t v = static_cast<std::size_t>(nn) * h_s_elems[p]; v < wide; ++v)
                    si[v] = -1;
                h_s_dof[p] = table.size();
            }
        }, 16);
        mx_ndof = 0;
        for (int p = 0; p < n_spaces; ++p)
            mx_ndof = std::max(mx_ndof, h_s_dof[p]);

        gI.resize(mx_ndof * n_spaces);
        int *h_gI = gI.host_write();
        detail::parallel_for(static_cast<std::size_t>(n_spaces), [&](std::size_t p0, std::size_t p1, int)
        {
            for (std::size_t p = p0; p < p1; ++p)
            {
                int *dst = h_gI + static_cast<std::size_t>(mx_ndof) * p;
                std::copy(s2g.data() + wide * p, s2g.data() + wide * p + h_s_dof[p], dst);
                std::fill(dst + h_s_dof[p], dst + mx_ndof, -1);
            }
        }, 16);

        timer.lap("EnsembleSpace: subspace dofs");
        // ---- face-space numbering: first touch over (face, i)
        fI.resize(nb * mx_faces * n_spaces);
        int *h_fI = fI.host_write();
        const std::size_t fwide = static_cast<std::size_t>(nb) * mx_faces;
        std::vector<int> f2s(fwide * n_spaces); // face-space dof -> subspace dof, stride `fwide`
        int *h_s_fdof = s_fdof.host_write();
        detail::parallel_for(static_cast<std::size_t>(n_spaces), [&](std::size_t p0, std::size_t p1, int)
        {
            std::vector<int> fslot(mx_ndof);
            for (std::size_t p = p0; p < p1; ++p)
            {
                std::fill(fslot.begin(), fslot.begin() + h_s_dof[p], -1);
                int *mine = f2s.data() + fwide * p;
                int *fi = h_fI + fwide * p;
                int count = 0;
                for (int f = 0; f < h_s_faces[p]; ++f)
                {
                    const SubFace sf = sub_faces[face_off[p] + f];
                    const Edge *edge = mesh.edge(sf.edge);
                    const int g_el = edge->elements[sf.side];
                    const int s = edge->sides[sf.side];
                    const bool flip = (sf.side == 1 && edge->delta < 0);
                    const int *si = h_sI + static_cast<std::size_t>(nn) * (local_elem[g_el] + static_cast<std::size_t>(mx_elems) * p);
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
                            fslot[d] = count;
                            mine[count++] = d;
                        }
                        fi[i + nb * f] = fslot[d];
                    }
                }
                for (std::size_t v = static_cast<std::size_t>(nb) * h_s_faces[p]; v < fwide; ++v)
                    fi[v] = -1;
                h_s_fdof[p] = count;
            }
        }, 16);
        mx_fdof = 0;
        for (int p = 0; p < n_spaces; ++p)
            mx_fdof = std::max(mx_fdof, h_s_fdof[p]);

        pI.resize(mx_fdof * n_spaces);
        int *h_pI = pI.host_write();
        detail::parallel_for(static_cast<std::size_t>(n_spaces), [&](std::size_t p0, std::size_t p1, int)
        {
            for (std::size_t p = p0; p < p1; ++p)
            {
                int *dst = h_pI + static_cast<std::size_t>(mx_fdof) * p;
                std::copy(f2s.data() + fwide * p, f2s.data() + fwide * p + h_s_fdof[p], dst);
                std::fill(dst + h_s_fdof[p], dst + mx_fdof, -1);
            }
        }, 16);

        timer.lap("EnsembleSpace: face dofs");
        // ---- shared dof pairs, unique per (unordered subspace pair, dof of the lower-numbered subspace), in the order the
        // shared faces (global edge order) reach them
        // A key can only repeat among the faces of ONE subspace pair, so the candidates are split by their lower subspace into
        // ranges, one per thread; every range walks the shared faces in order (first occurrence wins, as in the serial loop)
        // with a set small enough to stay in cache, and the survivors are collected in face order afterwards.
        const std::size_t n_cand = shared_faces.size() * nb;
        std::vector<unsigned char> keep(n_cand, 0);
        const int CR = detail::chunk_count(static_cast<std::size_t>(n_spaces), 64);
        detail::parallel_for(static_cast<std::size_t>(n_spaces), [&](std::size_t lo0, std::size_t lo1, int)
        {
            KeySet seen(n_cand / static_cast<std::size_t>(CR) + 64);
            for (std::size_t t = 0; t < shared_faces.size(); ++t)
            {
                const auto &sfc = shared_faces[t];
                const int S0 = sfc[0], S1 = sfc[1], f0 = sfc[2], f1 = sfc[3];
                const std::uint64_t lo = std::min(S0, S1), hi = std::max(S0, S1);
                if (lo < lo0 || lo >= lo1)
                    continue;
                const std::uint64_t pair_id = lo + static_cast<std::uint64_t>(n_spaces) * hi;
                for (int i = 0; i < nb; ++i)
                {
                    const int j = S0 < S1 ? h_fI[i + nb * (f0 + static_cast<std::size_t>(mx_faces) * S0)]
                                          : h_fI[i + nb * (f1 + static_cast<std::size_t>(mx_faces) * S1)];
                    keep[t * nb + i] = seen.insert(pair_id * static_cast<std::uint64_t>(mx_fdof + 1) + static_cast<std::uint64_t>(j)) ? 1 : 0;
                }
            }
        }, 64);
        std::vector<std::array<int, 4>> pairs;
        pairs.reserve(n_cand);
        for (std::size_t t = 0; t < shared_faces.size(); ++t)
        {
            const auto &sfc = shared_faces[t];
            const int S0 = sfc[0], S1 = sfc[1], f0 = sfc[2], f1 = sfc[3];
            for (int i = 0; i < nb; ++i)
                if (keep[t * nb + i])
                    pairs.push_back({S0, S1, h_fI[i + nb * (f0 + static_cast<std::size_t>(mx_faces) * S0)],
                                     h_fI[i + nb * (f1 + static_cast<std::size_t>(mx_faces) * S1)]});
        }

        n_shared_dofs = static_cast<int>(pairs.size());
        cmap.resize(4 * n_shared_dofs);
        int *h_cmap = cmap.host_write();
        for (int k = 0; k < n_shared_dofs; ++k)
            for (int c = 0; c < 4; ++c)
                h_cmap[c + 4 * k] = pairs[k][c];
        timer.lap("EnsembleSpace: shared pairs");
    }
} // namespace cuddh
