// Text mesh files, uniform refinement and a Morton partitioner (see meshio.hpp).
#include "cuddh/meshio.hpp"

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <numeric>

#include "cuddh/error.hpp"

namespace cuddh
{
    QuadMeshData read_mesh_files(const std::string &dir)
    {
        auto open = [&](const char *name)
        {
            std::ifstream f(dir + "/" + name);
            if (!f)
            {
                const std::string err = "load_mesh error: cannot open file: " + dir + "/" + name;
                cuddh_error(err.c_str());
            }
            return f;
        };
        int n_pts = 0, n_elem = 0;
        {
            std::ifstream info = open("info.txt");
            info >> n_pts >> n_elem;
            if (!info || n_pts < 4 || n_elem < 1)
                cuddh_error("load_mesh error: info.txt must hold the vertex and element counts.");
        }
        QuadMeshData m;
        m.xy.resize(static_cast<std::size_t>(2) * n_pts);
        m.elems.resize(static_cast<std::size_t>(4) * n_elem);
        {
            std::ifstream coo = open("coordinates.txt");
            for (double &v : m.xy)
                coo >> v;
            if (!coo)
                cuddh_error("load_mesh error: coordinates.txt is shorter than info.txt says.");
        }
        {
            std::ifstream el = open("elements.txt");
            for (int &v : m.elems)
            {
                el >> v;
                if (el && (v < 0 || v >= n_pts))
                    cuddh_error("load_mesh error: elements.txt names a vertex that does not exist.");
            }
            if (!el)
                cuddh_error("load_mesh error: elements.txt is shorter than info.txt says.");
        }
        return m;
    }

    Mesh2D load_mesh(const std::string &dir)
    {
        const QuadMeshData m = read_mesh_files(dir);
        return Mesh2D::from_vertices(m.n_pts(), m.xy.data(), m.n_elem(), m.elems.data());
    }

    QuadMeshData mesh_data(const Mesh2D &mesh)
    {
        QuadMeshData m;
        m.xy.resize(static_cast<std::size_t>(2) * mesh.n_nodes());
        m.elems.resize(static_cast<std::size_t>(4) * mesh.n_elem());
        for (int k = 0; k < mesh.n_nodes(); ++k)
        {
            m.xy[2 * k] = mesh.node(k).x[0];
            m.xy[2 * k + 1] = mesh.node(k).x[1];
        }
        for (int el = 0; el < mesh.n_elem(); ++el)
            for (int c = 0; c < 4; ++c)
                m.elems[4 * el + c] = mesh.element(el)->nodes[c];
        return m;
    }

    QuadMeshData refine_quads(const QuadMeshData &in, int times)
    {
        QuadMeshData cur = in;
        for (int round = 0; round < times; ++round)
        {
            const std::int64_t n_pts = cur.n_pts();
            const int n_elem = cur.n_elem();
            // side s of an element joins corner s and corner s + 1
            std::vector<std::int64_t> key(static_cast<std::size_t>(4) * n_elem);
            for (int el = 0; el < n_elem; ++el)
                for (int s = 0; s < 4; ++s)
                {
                    const std::int64_t a = cur.elems[4 * el + s], b = cur.elems[4 * el + (s + 1) % 4];
                    key[4 * static_cast<std::size_t>(el) + s] = std::min(a, b) * (n_pts + 1) + std::max(a, b);
                }
            std::vector<std::int64_t> uniq(key);
            std::sort(uniq.begin(), uniq.end());
            uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
            const std::int64_t n_mid = static_cast<std::int64_t>(uniq.size());
            if (n_pts + n_mid + n_elem > 2147483647LL || static_cast<std::int64_t>(n_elem) * 4 > 536870911LL)
                cuddh_error("refine_quads error: the refined mesh does not fit 32-bit indices.");

            QuadMeshData next;
            next.xy.resize(static_cast<std::size_t>(2) * (n_pts + n_mid + n_elem));
            std::copy(cur.xy.begin(), cur.xy.end(), next.xy.begin());
            for (std::int64_t m = 0; m < n_mid; ++m)
            {
                const std::int64_t a = uniq[m] / (n_pts + 1), b = uniq[m] % (n_pts + 1);
                next.xy[2 * (n_pts + m)] = 0.5 * (cur.xy[2 * a] + cur.xy[2 * b]);
                next.xy[2 * (n_pts + m) + 1] = 0.5 * (cur.xy[2 * a + 1] + cur.xy[2 * b + 1]);
            }
            next.elems.resize(static_cast<std::size_t>(16) * n_elem);
            for (int el = 0; el < n_elem; ++el)
            {
                const int *c = cur.elems.data() + 4 * el;
                int mid[4];
                double cx = 0.0, cy = 0.0;
                for (int s = 0; s < 4; ++s)
                {
                    const auto it = std::lower_bound(uniq.begin(), uniq.end(), key[4 * static_cast<std::size_t>(el) + s]);
                    mid[s] = static_cast<int>(n_pts + (it - uniq.begin()));
                    cx += cur.xy[2 * c[s]];
                    cy += cur.xy[2 * c[s] + 1];
                }
                const int cen = static_cast<int>(n_pts + n_mid + el);
                next.xy[2 * static_cast<std::size_t>(cen)] = cx / 4.0;
                next.xy[2 * static_cast<std::size_t>(cen) + 1] = cy / 4.0;
                const int child[4][4] = {{c[0], mid[0], cen, mid[3]}, {mid[0], c[1], mid[1], cen}, {cen, mid[1], c[2], mid[2]}, {mid[3], cen, mid[2], c[3]}};
                for (int k = 0; k < 4; ++k)
                    for (int v = 0; v < 4; ++v)
                        next.elems[16 * static_cast<std::size_t>(el) + 4 * k + v] = child[k][v];
            }
            cur.xy.swap(next.xy);
            cur.elems.swap(next.elems);
        }
        return cur;
    }

    namespace
    {
        inline std::uint64_t spread(std::uint32_t v)
        {
            std::uint64_t x = v;
            x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
            x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
            x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
            x = (x | (x << 2)) & 0x3333333333333333ull;
            x = (x | (x << 1)) & 0x5555555555555555ull;
            return x;
        }
    } // namespace

    std::vector<int> partition_elements(const Mesh2D &mesh, int n_parts)
    {
        const int nel = mesh.n_elem();
        if (n_parts < 1 || n_parts > nel)
            cuddh_error("partition_elements error: need 1 <= n_parts <= number of elements.");
        std::vector<double> cx(nel), cy(nel);
        double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
        const double mid[2] = {0.0, 0.0};
        for (int el = 0; el < nel; ++el)
        {
            double c[2];
            mesh.element(el)->physical_coordinates(mid, c);
            cx[el] = c[0];
            cy[el] = c[1];
            lo[0] = std::min(lo[0], c[0]);
            hi[0] = std::max(hi[0], c[0]);
            lo[1] = std::min(lo[1], c[1]);
            hi[1] = std::max(hi[1], c[1]);
        }
        const double sx = hi[0] > lo[0] ? 65535.0 / (hi[0] - lo[0]) : 0.0, sy = hi[1] > lo[1] ? 65535.0 / (hi[1] - lo[1]) : 0.0;
        std::vector<std::uint64_t> code(nel);
        for (int el = 0; el < nel; ++el)
        {
            const auto qx = static_cast<std::uint32_t>((cx[el] - lo[0]) * sx + 0.5), qy = static_cast<std::uint32_t>((cy[el] - lo[1]) * sy + 0.5);
            code[el] = spread(qx) | (spread(qy) << 1);
        }
        std::vector<int> order(nel);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return code[a] < code[b]; });
        std::vector<int> labels(nel);
        for (int r = 0; r < nel; ++r)
            labels[order[r]] = static_cast<int>(static_cast<std::int64_t>(r) * n_parts / nel);
        return labels;
    }
} // namespace cuddh
