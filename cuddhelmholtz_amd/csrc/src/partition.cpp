// Element partition + halo lists of the multi-device operator apply (partition.hpp).
#include "cuddh/partition.hpp"

#include <algorithm>
#include <cstdint>
#include <unordered_map>

#include "cuddh/error.hpp"
#include "cuddh/meshio.hpp"

namespace cuddh
{
    HelmholtzPartition HelmholtzPartition::build(const Mesh2D &gmesh, const Basis &basis, const H1Space &gfem, const FaceSpace &gfs, int rank,
                                                 int world)
    {
        if (rank < 0 || rank >= world)
            cuddh_error("HelmholtzPartition error: rank out of range.");
        const int n_elem = gmesh.n_elem(), ndof = gfem.size(), nb = basis.size(), nn = nb * nb;
        if (world > n_elem)
            cuddh_error("HelmholtzPartition error: more ranks than elements.");
        HelmholtzPartition p;
        p.rank = rank;
        p.world = world;
        p.ndof_global = ndof;

        const std::vector<int> elem_rank = partition_elements(gmesh, world);
        const int *I = gfem.global_indices(MemorySpace::HOST); // (nb, nb, n_elem): column e = the dofs of element e
        // owner of a dof = the lowest rank among the elements touching it
        std::vector<int> owner(ndof, world);
        for (int e = 0; e < n_elem; ++e)
            for (int k = 0; k < nn; ++k)
            {
                int &o = owner[I[k + static_cast<std::size_t>(nn) * e]];
                o = std::min(o, elem_rank[e]);
            }
        for (int e = 0; e < n_elem; ++e)
            if (elem_rank[e] == rank)
                p.my_elems.push_back(e);
        const int n_my = static_cast<int>(p.my_elems.size());

        // ---- the sub-mesh: this rank's elements, vertices renumbered in increasing global id
        const QuadMeshData g = mesh_data(gmesh);
        const int n_pts = g.n_pts();
        std::vector<int> verts;
        verts.reserve(static_cast<std::size_t>(4) * n_my);
        for (const int e : p.my_elems)
            for (int c = 0; c < 4; ++c)
                verts.push_back(g.elems[4 * static_cast<std::size_t>(e) + c]);
        std::sort(verts.begin(), verts.end());
        verts.erase(std::unique(verts.begin(), verts.end()), verts.end());
        std::vector<int> remap(n_pts, -1);
        for (std::size_t i = 0; i < verts.size(); ++i)
            remap[verts[i]] = static_cast<int>(i);
        std::vector<double> lxy(2 * verts.size());
        for (std::size_t i = 0; i < verts.size(); ++i)
        {
            lxy[2 * i] = g.xy[2 * static_cast<std::size_t>(verts[i])];
            lxy[2 * i + 1] = g.xy[2 * static_cast<std::size_t>(verts[i]) + 1];
        }
        std::vector<int> lel(static_cast<std::size_t>(4) * n_my);
        for (int le = 0; le < n_my; ++le)
            for (int c = 0; c < 4; ++c)
                lel[4 * static_cast<std::size_t>(le) + c] = remap[g.elems[4 * static_cast<std::size_t>(p.my_elems[le]) + c]];
        p.mesh.reset(new Mesh2D(Mesh2D::from_vertices(static_cast<int>(verts.size()), lxy.data(), n_my, lel.data())));
        p.fem.reset(new H1Space(*p.mesh, basis));
        p.n_loc = p.fem->size();

        // ---- local -> global dofs through the element index maps (element le of the sub-mesh IS element my_elems[le], same corners)
        const int *Il = p.fem->global_indices(MemorySpace::HOST);
        p.l2g.assign(p.n_loc, -1);
        for (int le = 0; le < n_my; ++le)
            for (int k = 0; k < nn; ++k)
            {
                const int l = Il[k + static_cast<std::size_t>(nn) * le], gd = I[k + static_cast<std::size_t>(nn) * p.my_elems[le]];
                if (p.l2g[l] >= 0 && p.l2g[l] != gd)
                    cuddh_error("HelmholtzPartition error: local and global numberings are inconsistent.");
                p.l2g[l] = gd;
            }
        {
            std::vector<int> chk(p.l2g);
            std::sort(chk.begin(), chk.end());
            if (chk.empty() || chk.front() < 0 || std::adjacent_find(chk.begin(), chk.end()) != chk.end())
                cuddh_error("HelmholtzPartition error: the local -> global dof map is not injective.");
        }

        // ---- physical boundary faces of the sub-mesh: its boundary edges whose end points are a boundary edge of the whole mesh
        auto key = [n_pts](int a, int b) { return static_cast<std::int64_t>(std::min(a, b)) * n_pts + std::max(a, b); };
        std::vector<std::int64_t> gkeys;
        for (int e = 0; e < gmesh.n_edges(); ++e)
            if (gmesh.edge(e)->type == FaceType::BOUNDARY)
                gkeys.push_back(key(gmesh.edge(e)->nodes[0], gmesh.edge(e)->nodes[1]));
        std::sort(gkeys.begin(), gkeys.end());
        for (int e = 0; e < p.mesh->n_edges(); ++e)
        {
            const Edge *ed = p.mesh->edge(e);
            if (ed->type != FaceType::BOUNDARY)
                continue;
            if (std::binary_search(gkeys.begin(), gkeys.end(), key(verts[ed->nodes[0]], verts[ed->nodes[1]])))
                p.faces.push_back(e);
        }
        p.fs.reset(new FaceSpace(*p.fem, static_cast<int>(p.faces.size()), p.faces.data()));
        {
            std::vector<int> g2f(ndof, -1);
            const int *gproj = gfs.global_indices(MemorySpace::HOST);
            for (int i = 0; i < gfs.size(); ++i)
                g2f[gproj[i]] = i;
            const int *lproj = p.fs->global_indices(MemorySpace::HOST);
            p.face_l2g.resize(p.fs->size());
            for (int i = 0; i < p.fs->size(); ++i)
            {
                p.face_l2g[i] = g2f[p.l2g[lproj[i]]];
                if (p.face_l2g[i] < 0)
                    cuddh_error("HelmholtzPartition error: a local boundary-face dof is not in the global FaceSpace.");
            }
        }

        // ---- ownership and the two exchanges
        std::vector<int> g2l(ndof, -1);
        for (int l = 0; l < p.n_loc; ++l)
        {
            g2l[p.l2g[l]] = l;
            (owner[p.l2g[l]] == rank ? p.owned : p.halo).push_back(l);
        }
        // my halo dofs, by owner, ordered by global id
        for (const int l : p.halo)
            p.halo_from[owner[p.l2g[l]]].push_back(l);
        for (auto &kv : p.halo_from)
            std::sort(kv.second.begin(), kv.second.end(), [&](int a, int b) { return p.l2g[a] < p.l2g[b]; });
        // my owned dofs that rank s holds (s touches them through one of ITS elements), ordered by global id: the mirror image of
        // s's halo_from[rank] list
        std::vector<int> stamp(ndof, -1);
        for (int e = 0; e < n_elem; ++e)
        {
            const int s = elem_rank[e];
            if (s == rank)
                continue;
            for (int k = 0; k < nn; ++k)
            {
                const int gd = I[k + static_cast<std::size_t>(nn) * e];
                if (owner[gd] == rank && stamp[gd] != s)
                {
                    // a dof is held by at most a few ranks: `stamp` de-duplicates runs of the same holder, the sort + unique below the rest
                    stamp[gd] = s;
                    p.own_to[s].push_back(gd);
                }
            }
        }
        for (auto &kv : p.own_to)
        {
            std::sort(kv.second.begin(), kv.second.end());
            kv.second.erase(std::unique(kv.second.begin(), kv.second.end()), kv.second.end());
            for (int &gd : kv.second)
                gd = g2l[gd]; // global -> local, order kept
        }
        return p;
    }
} // namespace cuddh
