// Mesh construction and metric caches.
#include "cuddh/mesh.hpp"

#include "cuddh/error.hpp"
#include "cuddh/launch.hpp"
#include "cuddh/parallel.hpp"
#include "cuddh_hip.h"

#include <algorithm>
#include <cstdint>
#include <limits>
#include <new>

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
            QuadElement q(X);
            q.id = el;
            for (int i = 0; i < 4; ++i)
                q.nodes[i] = c[i];
            mesh._elements[el] = q;
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
                    StraightEdge e(x_ + 2 * v0, x_ + 2 * v1, s);
                    e.id = id;
                    e.type = FaceType::BOUNDARY;
                    e.nodes[0] = v0;
                    e.nodes[1] = v1;
                    e.elements[0] = el;
                    e.sides[0] = s;
                    e.delta = 1;
                    mesh._edges.push_back(e);
                    seen.emplace(key, id);
                }
                else
                {
                    Edge &e = mesh._edges[it->second];
                    e.type = FaceType::INTERIOR;
                    e.elements[1] = el;
                    e.sides[1] = s;
                    // same direction iff this element's start vertex is the first element's start vertex
                    e.delta = (v0 == e.nodes[0]) ? 1 : -1;
                    mesh._nodes[v0].type = NodeType::INTERIOR;
                    mesh._nodes[v1].type = NodeType::INTERIOR;
                }
            }
        }
        mesh.classify();
        return mesh;
    }

    void Mesh2D::classify()
    {
        for (const auto &e : _edges)
            (e.type == FaceType::BOUNDARY ? _boundary_edges : _interior_edges).push_back(e.id);
        for (const auto &nd : _nodes)
            (nd.type == NodeType::BOUNDARY ? _boundary_nodes : _interior_nodes).push_back(nd.id);
    }

    // The mesh from_vertices() builds for the lattice of (nx+1) x (ny+1) vertices (vertex id i + (nx+1) j, element id
    // i + nx j, corners sw, se, ne, nw; reference source/Mesh2D.cpp:138-171), written down in closed form: what the
    // edge hash of from_vertices discovers is known in advance on a lattice, so rows are filled in parallel.
    //   * edge ids in first-seen order over (element, side 0..3): element (i, j) is the first to see its bottom edge iff
    //     j == 0, its right and top edges always, its left edge iff i == 0;
    //   * the right edge of (i, j) is side 3 of (i+1, j), the top edge side 0 of (i, j+1), both traversed in the same
    //     direction (delta = +1: uniform_rect never produces a reversed edge);
    //   * a vertex is INTERIOR iff an interior edge ends in it, i.e. unless it is one of the four corners of the lattice
    //     (or, on a one-element-wide lattice, lies on a side no interior edge reaches);
    //   * a vertex lists the elements around it in increasing element id: sw (as its corner 2), se (3), nw (1), ne (0).
    // tests/test_host_numbering.py compares the result with the oracle's restatement of from_vertices.
    Mesh2D Mesh2D::uniform_rect(int nx, double ax, double bx, int ny, double ay, double by)
    {
        if (nx < 1 || ny < 1)
            cuddh_error("Mesh2D::uniform_rect error: nx and ny must be positive.");
        const int npx = nx + 1, npy = ny + 1;
        const double dx = (bx - ax) / nx, dy = (by - ay) / ny;
        auto vx = [&](int i) { return ax + dx * i; };
        auto vy = [&](int j) { return ay + dy * j; };

        detail::PhaseTimer timer;
        Mesh2D mesh;
        // raw slots: every node, element and edge is constructed in place by the row loops below, exactly once
        Node *nodes = mesh._nodes.claim(static_cast<std::size_t>(npx) * npy);
        QuadElement *elements = mesh._elements.claim(static_cast<std::size_t>(nx) * ny);
        const std::size_t row0 = static_cast<std::size_t>(3) * nx + 1, rowk = static_cast<std::size_t>(2) * nx + 1;
        StraightEdge *edges = mesh._edges.claim(row0 + rowk * (ny - 1));

        detail::parallel_for(static_cast<std::size_t>(npy), [&](std::size_t j0, std::size_t j1, int)
        {
            for (int j = static_cast<int>(j0); j < static_cast<int>(j1); ++j)
                for (int i = 0; i < npx; ++i)
                {
                    Node &nd = *new (nodes + static_cast<std::size_t>(i) + static_cast<std::size_t>(npx) * j) Node();
                    nd.id = i + npx * j;
                    nd.x[0] = vx(i);
                    nd.x[1] = vy(j);
                    const bool vertical_interior = i > 0 && i < nx;   // an interior vertical edge ends here
                    const bool horizontal_interior = j > 0 && j < ny; // an interior horizontal edge ends here
                    nd.type = (vertical_interior || horizontal_interior) ? NodeType::INTERIOR : NodeType::BOUNDARY;
                    nd.connected_elements.reserve(4);
                    if (i > 0 && j > 0)
                        nd.connected_elements.push_back({2, (i - 1) + nx * (j - 1)});
                    if (i < nx && j > 0)
                        nd.connected_elements.push_back({3, i + nx * (j - 1)});
                    if (i > 0 && j < ny)
                        nd.connected_elements.push_back({1, (i - 1) + nx * j});
                    if (i < nx && j < ny)
                        nd.connected_elements.push_back({0, i + nx * j});
                }
        }, 8);
        timer.lap("mesh: nodes");

        detail::parallel_for(static_cast<std::size_t>(ny), [&](std::size_t j0, std::size_t j1, int)
        {
            for (int j = static_cast<int>(j0); j < static_cast<int>(j1); ++j)
            {
                // first edge id of row j, then 3 (row 0) or 2 new edges per element, plus the left edge of element 0
                const std::size_t base = j == 0 ? 0 : row0 + rowk * (j - 1);
                const int per = j == 0 ? 3 : 2;
                for (int i = 0; i < nx; ++i)
                {
                    const int el = i + nx * j;
                    const int sw = i + npx * j, se = sw + 1, ne = sw + 1 + npx, nw = sw + npx;
                    const double X[8] = {vx(i), vy(j), vx(i + 1), vy(j), vx(i + 1), vy(j + 1), vx(i), vy(j + 1)};
                    QuadElement q(X);
                    q.id = el;
                    q.nodes[0] = sw;
                    q.nodes[1] = se;
                    q.nodes[2] = ne;
                    q.nodes[3] = nw;
                    new (elements + el) QuadElement(q);

                    std::size_t id = base + static_cast<std::size_t>(per) * i + (i > 0 ? 1 : 0);
                    auto put = [&](int side, int v0, int v1, const double *x0, const double *x1, int el1, int side1)
                    {
                        StraightEdge e(x0, x1, side);
                        e.id = static_cast<int>(id);
                        e.nodes[0] = v0;
                        e.nodes[1] = v1;
                        e.elements[0] = el;
                        e.sides[0] = side;
                        e.delta = 1;
                        if (el1 >= 0)
                        {
                            e.type = FaceType::INTERIOR;
                            e.elements[1] = el1;
                            e.sides[1] = side1;
                        }
                        else
                            e.type = FaceType::BOUNDARY;
                        new (edges + id++) StraightEdge(e);
                    };
                    if (j == 0)
                        put(0, sw, se, X + 0, X + 2, -1, -1);                                  // bottom: c0 -> c1
                    put(1, se, ne, X + 2, X + 4, i + 1 < nx ? el + 1 : -1, 3);                 // right:  c1 -> c2
                    put(2, nw, ne, X + 6, X + 4, j + 1 < ny ? el + nx : -1, 0);                // top:    c3 -> c2
                    if (i == 0)
                        put(3, sw, nw, X + 0, X + 6, -1, -1);                                  // left:   c0 -> c3
                }
            }
        }, 8);
        timer.lap("mesh: elements and edges");
        mesh.classify();
        timer.lap("mesh: boundary/interior lists");
        return mesh;
    }

    double Mesh2D::min_h() const
    {
        double h = std::numeric_limits<double>::infinity();
        for (const auto &e : _edges)
            h = std::min(h, e.length());
        return h;
    }

    double Mesh2D::max_h() const
    {
        double h = -1.0;
        for (const auto &e : _edges)
            h = std::max(h, e.length());
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
            detail::parallel_for(static_cast<std::size_t>(nel), [&](std::size_t e0, std::size_t e1, int)
            {
                for (int el = static_cast<int>(e0); el < static_cast<int>(e1); ++el)
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
            });
        }
    } // namespace

    void Mesh2D::ElementMetricCollection::ensure_corners() const
    {
        const int m = quad.size(), nel = mesh.n_elem();
        if (corners.size() == 0)
        {
            corners.resize(8 * nel);
            double *c = corners.host_write();
            detail::parallel_for(static_cast<std::size_t>(nel), [&](std::size_t e0, std::size_t e1, int)
            {
                for (std::size_t el = e0; el < e1; ++el)
                {
                    const QuadElement &q = mesh._elements[el];
                    for (int k = 0; k < 4; ++k)
                    {
                        c[8 * el + 2 * k] = q.corner(k)[0];
                        c[8 * el + 2 * k + 1] = q.corner(k)[1];
                    }
                }
            });
            points.resize(m);
            double *p = points.host_write();
            for (int i = 0; i < m; ++i)
                p[i] = quad.x(i);
        }
    }

    const double *Mesh2D::ElementMetricCollection::corner_coordinates_device() const
    {
        ensure_corners();
        return corners.device_read();
    }

    const double *Mesh2D::ElementMetricCollection::rule_nodes_device() const
    {
        ensure_corners();
        return points.device_read();
    }

    void Mesh2D::ElementMetricCollection::on_device(host_device_dvec &out, int dim, int which) const
    {
        const int m = quad.size(), nel = mesh.n_elem();
        ensure_corners();
        out.resize(dim * m * m * nel);
        double *d = out.device_write();
        detail::check_hip(cuddh_hip_element_metrics(nel, m, corners.device_read(), points.device_read(), which == 0 ? d : nullptr,
                                                    which == 1 ? d : nullptr, which == 2 ? d : nullptr, stream()),
                          "element metrics");
    }

    const double *Mesh2D::ElementMetricCollection::jacobians(MemorySpace m) const
    {
        if (J.size() == 0 && m == MemorySpace::DEVICE)
            on_device(J, 4, 0);
        if (J.size() == 0)
            tabulate_elements(J, 4, mesh, quad, [](double *o, const Element *e, const double *xi) { e->jacobian(xi, o); });
        return J.read(m);
    }

    const double *Mesh2D::ElementMetricCollection::measures(MemorySpace m) const
    {
        if (detJ.size() == 0 && m == MemorySpace::DEVICE)
            on_device(detJ, 1, 1);
        if (detJ.size() == 0)
            tabulate_elements(detJ, 1, mesh, quad, [](double *o, const Element *e, const double *xi) { *o = e->measure(xi); });
        return detJ.read(m);
    }

    const double *Mesh2D::ElementMetricCollection::physical_coordinates(MemorySpace m) const
    {
        if (x.size() == 0 && m == MemorySpace::DEVICE)
            on_device(x, 2, 2);
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
