// Conforming mesh of straight-sided quadrilaterals and its cached metric arrays.
// Contract: reference include/Mesh2D.hpp:17-299.  Numbering conventions the
// index maps depend on (reference source/Mesh2D.cpp:16-17,70-115,145-168):
// local sides {0: c0->c1, 1: c1->c2, 2: c3->c2, 3: c0->c3}; edge ids in
// first-seen order over (element, side); `uniform_rect` element id i + nx*j,
// vertex id i + (nx+1)*j.  Edge lookup uses 64-bit keys, so meshes larger than
// 214x214 elements are safe (the reference's 32-bit key overflows there,
// source/Mesh2D.cpp:64-67).  `uniform_rect` builds the same mesh in closed form (no hashing, parallel over rows).
#ifndef CUDDH_AMD_MESH_HPP
#define CUDDH_AMD_MESH_HPP

#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "geometry.hpp"
#include "memory.hpp"
#include "quadrature.hpp"
#include "tensor.hpp"

namespace cuddh
{
    namespace detail
    {
        /// Contiguous store of mesh entities whose slots can be constructed in parallel: std::vector value-initialises on one
        /// thread (0.4 GB of nodes, edges and elements at 1024^2: 0.08 s of a 0.1 s mesh).  `claim(n)` allocates n raw slots that
        /// the caller constructs with placement new (every slot exactly once); push_back grows like a vector.
        template <typename T>
        class EntityStore
        {
        public:
            EntityStore() = default;
            EntityStore(const EntityStore &) = delete;
            EntityStore &operator=(const EntityStore &) = delete;
            EntityStore(EntityStore &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
            EntityStore &operator=(EntityStore &&o) noexcept
            {
                if (this != &o)
                {
                    release();
                    p = o.p;
                    n = o.n;
                    cap = o.cap;
                    o.p = nullptr;
                    o.n = o.cap = 0;
                }
                return *this;
            }
            ~EntityStore() { release(); }

            std::size_t size() const { return n; }
            T &operator[](std::size_t i) { return p[i]; }
            const T &operator[](std::size_t i) const { return p[i]; }
            T *begin() { return p; }
            T *end() { return p + n; }
            const T *begin() const { return p; }
            const T *end() const { return p + n; }

            /// n raw slots (the store must be empty); the caller placement-constructs each of them
            T *claim(std::size_t count)
            {
                release();
                p = static_cast<T *>(::operator new(count * sizeof(T)));
                n = cap = count;
                return p;
            }
            /// value-constructed slots, on this thread (the generic from_vertices path)
            void resize(std::size_t count)
            {
                T *q = claim(count);
                for (std::size_t i = 0; i < count; ++i)
                    new (q + i) T();
            }
            void push_back(const T &v)
            {
                if (n == cap)
                {
                    const std::size_t ncap = cap ? 2 * cap : 1024;
                    T *q = static_cast<T *>(::operator new(ncap * sizeof(T)));
                    for (std::size_t i = 0; i < n; ++i)
                    {
                        new (q + i) T(std::move(p[i]));
                        p[i].~T();
                    }
                    ::operator delete(p);
                    p = q;
                    cap = ncap;
                }
                new (p + n++) T(v);
            }

        private:
            void release()
            {
                for (std::size_t i = 0; i < n; ++i)
                    p[i].~T();
                ::operator delete(p);
                p = nullptr;
                n = cap = 0;
            }
            T *p = nullptr;
            std::size_t n = 0, cap = 0;
        };
    } // namespace detail

    class Mesh2D
    {
    public:
        /// Lazily evaluated per-element metric arrays on the tensor grid of a 1-D rule.
        class ElementMetricCollection
        {
        public:
            ElementMetricCollection(const Mesh2D &mesh, const QuadratureRule &quad) : mesh(mesh), quad(quad) {}

            /// shape (2, 2, n, n, n_elem): J(a, b, i, j, el) = d x_a / d xi_b
            const double *jacobians(MemorySpace m) const;
            /// shape (n, n, n_elem)
            const double *measures(MemorySpace m) const;
            /// shape (2, n, n, n_elem)
            const double *physical_coordinates(MemorySpace m) const;

            /// DEVICE inputs of the on-device evaluation, for kernels that evaluate the bilinear map themselves instead of
            /// reading a table: corner coordinates (2, 4, n_elem), counter-clockwise, and the rule's n nodes
            const double *corner_coordinates_device() const;
            const double *rule_nodes_device() const;

        private:
            void ensure_corners() const;
            /// first request on the DEVICE: evaluated there from the elements' corners (no host table, no upload)
            void on_device(host_device_dvec &out, int dim, int which) const;

            const Mesh2D &mesh;
            QuadratureRule quad;
            mutable host_device_dvec J, detJ, x;
            mutable host_device_dvec corners, points; // (2, 4, n_elem) and the rule's nodes, device inputs of on_device()
        };

        /// Lazily evaluated per-edge metric arrays on a 1-D rule, for all edges of
        /// one type or for an explicit list of edges.
        class EdgeMetricCollection
        {
        public:
            EdgeMetricCollection(const Mesh2D &mesh, const FaceType edge_type, const QuadratureRule &quad);
            EdgeMetricCollection(const Mesh2D &mesh, int n_faces, const int *faces, const QuadratureRule &quad);

            /// shape (n, n_edges)
            const double *measures(MemorySpace m) const;
            /// shape (2, n, n_edges)
            const double *physical_coordinates(MemorySpace m) const;
            /// shape (2, n, n_edges)
            const double *normals(MemorySpace m) const;

        private:
            template <typename Eval>
            void fill(host_device_dvec &out, int dim, Eval eval) const;

            const Mesh2D &mesh;
            QuadratureRule quad;
            std::vector<int> edge_ids; // global ids of the edges covered, in output order
            mutable host_device_dvec detJ, x, n;
        };

        Mesh2D() = default;
        ~Mesh2D() = default;
        Mesh2D(const Mesh2D &) = delete;
        Mesh2D &operator=(const Mesh2D &) = delete;
        Mesh2D(Mesh2D &&) = default;
        Mesh2D &operator=(Mesh2D &&) = default;

        int n_elem() const { return static_cast<int>(_elements.size()); }
        int n_edges() const { return static_cast<int>(_edges.size()); }
        int n_edges(FaceType type) const
        {
            return static_cast<int>(type == FaceType::BOUNDARY ? _boundary_edges.size() : _interior_edges.size());
        }
        int n_nodes() const { return static_cast<int>(_nodes.size()); }
        int n_nodes(NodeType type) const
        {
            return static_cast<int>(type == NodeType::BOUNDARY ? _boundary_nodes.size() : _interior_nodes.size());
        }

        int max_element_order() const { return 1; }
        int min_element_order() const { return 1; }

        double min_h() const;
        double max_h() const;

        const Node &node(int i) const { return _nodes[i]; }
        const Node &node(int i, NodeType type) const
        {
            return _nodes[type == NodeType::BOUNDARY ? _boundary_nodes[i] : _interior_nodes[i]];
        }

        const Edge *edge(int i) const { return &_edges[i]; }
        const Edge *edge(int i, FaceType type) const
        {
            return &_edges[type == FaceType::BOUNDARY ? _boundary_edges[i] : _interior_edges[i]];
        }

        ivec boundary_edges() const;

        const Element *element(int el) const { return &_elements[el]; }

        const ElementMetricCollection &element_metrics(const QuadratureRule &quad) const;
        const EdgeMetricCollection &edge_metrics(const QuadratureRule &quad, FaceType edge_type) const;

        /// x: (2, nx) vertex coordinates; elems: (4, nel) corner vertex ids, counter-clockwise
        static Mesh2D from_vertices(int nx, const double *x, int nel, const int *elems);
        static Mesh2D uniform_rect(int nx, double ax, double bx, int ny, double ay, double by);

    private:
        /// boundary / interior lists from the node and edge types, in id order
        void classify();

        // contiguous stores (the reference keeps one heap object per edge and element, include/Mesh2D.hpp:283-285;
        // straight edges and bilinear quadrilaterals are the only kinds either code has)
        detail::EntityStore<Node> _nodes;
        detail::EntityStore<StraightEdge> _edges;
        detail::EntityStore<QuadElement> _elements;
        std::vector<int> _interior_nodes, _boundary_nodes;
        std::vector<int> _boundary_edges, _interior_edges;

        mutable std::unordered_map<std::string, std::unique_ptr<ElementMetricCollection>> elem_cache;
        mutable std::unordered_map<std::string, std::unique_ptr<EdgeMetricCollection>> edge_cache[2];
    };
} // namespace cuddh

#endif
