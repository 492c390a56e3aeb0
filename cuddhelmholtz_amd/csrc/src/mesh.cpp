// Mesh construction and metric caches.
#include "cuddh/mesh.hpp"

#include <algorithm>
#include <cstdint>
#include <limits>

namespace cuddh
{
    namespace
    {
        // local side s joins corners side_from[s] -> side_to[s]
        constexpr int side_from[4] = {0, 1, 3, 0};
        constexpr int side_to[4] = {1, 2, 2, 3};

        inline std::uint64_t pair_key(int a, int b)
        {
            const std::uint64_t lo = static_cast<std::uint32_t>(std::min(a, b));
            const std::uint64_t hi = static_cast<std::uint32_t>(std::max(a, b));
            return (hi << 32) | lo;
        }
    } // namespace

    Mesh2D Mesh2D::from_vertices(int nx, const double *x_, int nel, const int *elems_)
    {
        Mesh2D mesh;
        mesh._nodes.resize(nx);
        mesh._elements.resize(nel);

        for (int k = 0; k < nx; ++k)
        {
            Node &nd = mesh._nodes[k];
            nd.id = k;
            nd.type = NodeType::BOUNDARY; // promoted to INTERIOR when an interior edge touches it
            nd.x[0] = x_[2 * k];
            nd.x[1] = x_[2 * k + 1];
        }

        for (int el = 0; el < nel; ++el)
        {
            double X[8];
            const int *c = elems_ + 4 * el;
            for (int i = 0; i < 4; ++i)
            {
                X[2 * i] = x_[2 * c[i]];
                X[2 * i + 1] = x_[2 * c[i] + 1];
                mesh._nodes[c[i]].connected_elements.push_back({i, el});
            }
            auto q = std::make_unique<QuadElement>(X);
            q->id = el;
            for (int i = 0; i < 4; ++i)
                q->nodes[i] = c[i];
            mesh._elements[el] = std::move(q);
        }

        // edges, numbered in the order (element, side) first reaches them
        std::unordered_map<std::uint64_t, int> seen;
        seen.reserve(static_cast<std::size_t>(nel) * 2 + 16);
        for (int el = 0; el < nel; ++el)
        {
            const int *c = elems_ + 4 * el;
            for (int s = 0; s < 4; ++s)
            {
                const int v0 = c[side_from[s]], v1 = c[side_to[s]];
                const auto key = pair_key(v0, v1);
                auto it = seen.find(key);
                if (it == seen.end())
                {
                    const int id = static_cast<int>(mesh._edges.size());
                    auto e = std::make_unique<StraightEdge>(x_ + 2 * v0, x_ + 2 * v1, s);
                    e->id = id;
                    e->type = FaceType::BOUNDARY;
                    e->nodes[0] = v0;
                    e->nodes[1] = v1;
                    e->elements[0] = el;
                    e->sides[0] = s;
                    e->delta = 1;
                    mesh._edges.push_back(std::move(e));
                    seen.emplace(key, id);
                }
                else
                {
                    Edge *e = mesh._edges[it->second].get();
                    e->type = FaceType::INTERIOR;
                    e->elements[1] = el;
                    e->sides[1] = s;
                    // same direction iff this element's start vertex is the first element's start vertex
                    e->delta = (v0 == e->nodes[0]) ? 1 : -1;
                    mesh._nodes[v0].type = NodeType::INTERIOR;
                    mesh._nodes[v1].type = NodeType::INTERIOR;
                }
            }
        }

        for (const auto &e : mesh._edges)
            (e->type == FaceType::BOUNDARY ? mesh._boundary_edges : mesh._interior_edges).push_back(e->id);
        for (const auto &nd : mesh._nodes)
            (nd.type == NodeType::BOUNDARY ? mesh._boundary_nodes : mesh._interior_nodes).push_back(nd.id);

        return mesh;
    }

    Mesh2D Mesh2D::uniform_rect(int nx, double ax, double bx, int ny, double ay, double by)
    {
        const int npx = nx + 1, npy = ny + 1;
        std::vector<double> coo(static_cast<std::size_t>(2) * npx * npy);
        std::vector<int> quads(static_cast<std::size_t>(4) * nx * ny);

        const double dx = (bx - ax) / nx, dy = (by - ay) / ny;
        for (int j = 0; j < npy; ++j)
        {
            const double y = ay + dy * j;
            for (int i = 0; i < npx; ++i)
            {
                const std::size_t v = static_cast<std::size_t>(i) + static_cast<std::size_t>(npx) * j;
                coo[2 * v] = ax + dx * i;
                coo[2 * v + 1] = y;
            }
        }
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i)
            {
                int *c = quads.data() + 4 * (static_cast<std::size_t>(i) + static_cast<std::size_t>(nx) * j);
                const int sw = i + npx * j;
                c[0] = sw;
                c[1] = sw + 1;
                c[2] = sw + 1 + npx;
                c[3] = sw + npx;
            }
        return from_vertices(npx * npy, coo.data(), nx * ny, quads.data());
    }

    double Mesh2D::min_h() const
    {
        double h = std::numeric_limits<double>::infinity();
        for (const auto &e : _edges)
            h = std::min(h, e->length());
        return h;
    }

    double Mesh2D::max_h() const
    {
        double h = -1.0;
        for (const auto &e : _edges)
            h = std::max(h, e->length());
        return h;
    }

    ivec Mesh2D::boundary_edges() const
    {
        ivec out(static_cast<int>(_boundary_edges.size()));
        std::copy(_boundary_edges.begin(), _boundary_edges.end(), out.begin());
        return out;
    }

    const Mesh2D::ElementMetricCollection &Mesh2D::element_metrics(const QuadratureRule &quad) const
    {
        auto &slot = elem_cache[quad.name()];
        if (!slot)
            slot = std::make_unique<ElementMetricCollection>(*this, quad);
        return *slot;
    }

    const Mesh2D::EdgeMetricCollection &Mesh2D::edge_metrics(const QuadratureRule &quad, FaceType edge_type) const
    {
        auto &slot = edge_cache[edge_type == FaceType::INTERIOR ? 0 : 1][quad.name()];
        if (!slot)
            slot = std::make_unique<EdgeMetricCollection>(*this, edge_type, quad);
        return *slot;
    }

    // ------------------------------------------------------------ element metrics

    namespace
    {
        template <typename Eval>
        void tabulate_elements(host_device_dvec &out, int dim, const Mesh2D &mesh, const QuadratureRule &quad, Eval eval)
        {
            const int m = quad.size(), nel = mesh.n_elem();
            out.resize(dim * m * m * nel);
            double *dst = out.host_write();
            for (int el = 0; el < nel; ++el)
            {
                const Element *e = mesh.element(el);
                double *d = dst + static_cast<std::size_t>(dim) * m * m * el;
                for (int j = 0; j < m; ++j)
                    for (int i = 0; i < m; ++i)
                    {
                        const double xi[2] = {quad.x(i), quad.x(j)};
                        eval(d + dim * (i + m * j), e, xi);
                    }
            }
        }
    } // namespace

    const double *Mesh2D::ElementMetricCollection::jacobians(MemorySpace m) const
    {
        if (J.size() == 0)
            tabulate_elements(J, 4, mesh, quad, [](double *o, const Element *e, const double *xi) { e->jacobian(xi, o); });
        return J.read(m);
    }

    const double *Mesh2D::ElementMetricCollection::measures(MemorySpace m) const
    {
        if (detJ.size() == 0)
            tabulate_elements(detJ, 1, mesh, quad, [](double *o, const Element *e, const double *xi) { *o = e->measure(xi); });
        return detJ.read(m);
    }

    const double *Mesh2D::ElementMetricCollection::physical_coordinates(MemorySpace m) const
    {
        if (x.size() == 0)
            tabulate_elements(x, 2, mesh, quad, [](double *o, const Element *e, const double *xi) { e->physical_coordinates(xi, o); });
        return x.read(m);
    }

    // ------------------------------------------------------------ edge metrics

    Mesh2D::EdgeMetricCollection::EdgeMetricCollection(const Mesh2D &mesh_, const FaceType edge_type, const QuadratureRule &quad_)
        : mesh(mesh_), quad(quad_)
    {
        const int ne = mesh.n_edges(edge_type);
        edge_ids.resize(ne);
        for (int e = 0; e < ne; ++e)
            edge_ids[e] = mesh.edge(e, edge_type)->id;
    }

    Mesh2D::EdgeMetricCollection::EdgeMetricCollection(const Mesh2D &mesh_, int n_faces, const int *faces, const QuadratureRule &quad_)
        : mesh(mesh_), quad(quad_), edge_ids(faces, faces + n_faces)
    {
    }

    template <typename Eval>
    void Mesh2D::EdgeMetricCollection::fill(host_device_dvec &out, int dim, Eval eval) const
    {
        const int m = quad.size(), ne = static_cast<int>(edge_ids.size());
        out.resize(dim * m * ne);
        double *dst = out.host_write();
        for (int e = 0; e < ne; ++e)
        {
            const Edge *E = mesh.edge(edge_ids[e]);
            for (int i = 0; i < m; ++i)
                eval(dst + dim * (i + static_cast<std::size_t>(m) * e), E, quad.x(i));
        }
    }

    const double *Mesh2D::EdgeMetricCollection::measures(MemorySpace m) const
    {
        if (detJ.size() == 0)
            fill(detJ, 1, [](double *o, const Edge *E, double xi) { *o = E->measure(xi); });
        return detJ.read(m);
    }

    const double *Mesh2D::EdgeMetricCollection::physical_coordinates(MemorySpace m) const
    {
        if (x.size() == 0)
            fill(x, 2, [](double *o, const Edge *E, double xi) { E->physical_coordinates(xi, o); });
        return x.read(m);
    }

    const double *Mesh2D::EdgeMetricCollection::normals(MemorySpace m) const
    {
        if (n.size() == 0)
            fill(n, 2, [](double *o, const Edge *E, double xi) { E->normal(xi, o); });
        return n.read(m);
    }
} // namespace cuddh
