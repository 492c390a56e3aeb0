// Mesh entities: nodes, straight edges, bilinear quadrilateral elements.
// Contracts: reference include/Node.hpp:8-26, include/Edge.hpp:10-157,
// include/Element.hpp:10-120.  Conventions consumed by the index maps:
// corners counter-clockwise; Jacobian stored [x_xi, y_xi, x_eta, y_eta]
// (source/Element.cpp:21-27); edge reference measure = length / 2.
#ifndef CUDDH_AMD_GEOMETRY_HPP
#define CUDDH_AMD_GEOMETRY_HPP

#include <cmath>
#include <vector>

#include "tensor.hpp"

namespace cuddh
{
    enum class NodeType
    {
        INTERIOR,
        BOUNDARY
    };

    struct Node
    {
        struct element_info
        {
            int i;  ///< which corner (0..3) of the element this node is
            int id; ///< global element index
        };

        /// The elements around a node.  Same use as the reference's std::vector (include/Node.hpp:24: push_back, size,
        /// empty, indexing, iteration), but the first six entries live inside the node: a structured mesh of a million
        /// elements then costs no heap allocation per node (0.1 s of the 1024^2 set-up); longer lists spill to a vector.
        class element_list
        {
        public:
            void push_back(const element_info &e)
            {
                if (n < INLINE)
                    head[n] = e;
                else
                    tail.push_back(e);
                ++n;
            }
            std::size_t size() const { return static_cast<std::size_t>(n); }
            bool empty() const { return n == 0; }
            const element_info &operator[](std::size_t k) const { return k < INLINE ? head[k] : tail[k - INLINE]; }
            void reserve(std::size_t) {}

        private:
            static constexpr int INLINE = 6;
            element_info head[INLINE];
            std::vector<element_info> tail;
            int n = 0;
        };

        int id;
        NodeType type;
        double x[2];
        element_list connected_elements;
    };

    enum class FaceType
    {
        INTERIOR, ///< two elements meet here: elements[1], sides[1] are defined
        BOUNDARY  ///< physical boundary: only elements[0], sides[0] are defined
    };

    struct Edge
    {
        FaceType type;
        int id;          ///< global edge index
        int nodes[2];    ///< end points (global node indices)
        int elements[2]; ///< adjacent elements
        int sides[2];    ///< local side (0..3) of this edge in each adjacent element
        int delta;       ///< +1 if both elements traverse the edge in the same direction, -1 otherwise

        Edge() : type(FaceType::BOUNDARY), id(-1), nodes{-1, -1}, elements{-1, -1}, sides{-1, -1}, delta(1) {}
        virtual ~Edge() = default;

        /// unit normal pointing out of elements[0]
        virtual void normal(const double xi, double *n) const = 0;
        /// ds = measure(xi) d(xi), xi in [-1, 1]
        virtual double measure(const double xi) const = 0;
        virtual void physical_coordinates(const double xi, double *x) const = 0;
        virtual double length() const = 0;
    };

    struct StraightEdge : public Edge
    {
        StraightEdge() : a{0.0, 0.0}, t{0.0, 0.0}, nrm{0.0, 0.0}, len(0.0) {} // slot of a mesh's edge store, assigned later

        /// segment x0 -> x1; `side` is the local side index in the first element (fixes the normal's sign)
        StraightEdge(const double *x0, const double *x1, int side)
        {
            a[0] = x0[0];
            a[1] = x0[1];
            t[0] = x1[0] - x0[0];
            t[1] = x1[1] - x0[1];
            len = std::hypot(t[0], t[1]);
            const double s = (side == 2 || side == 3) ? -1.0 : 1.0;
            nrm[0] = s * t[1] / len;
            nrm[1] = -s * t[0] / len;
        }

        void normal(const double, double *n) const override
        {
            n[0] = nrm[0];
            n[1] = nrm[1];
        }

        double measure(const double) const override { return 0.5 * len; }

        void physical_coordinates(const double xi, double *x) const override
        {
            const double s = 0.5 * (xi + 1.0);
            x[0] = a[0] + s * t[0];
            x[1] = a[1] + s * t[1];
        }

        double length() const override { return len; }

    private:
        double a[2];   // start point
        double t[2];   // end - start
        double nrm[2]; // outward unit normal
        double len;
    };

    class Element
    {
    public:
        int id;
        int nodes[4];

        virtual ~Element() = default;
        virtual void physical_coordinates(const double *xi, double *x) const = 0;
        /// J = [dx/dxi, dy/dxi, dx/deta, dy/deta]
        virtual void jacobian(const double *xi, double *J) const = 0;
        virtual double measure(const double *xi) const
        {
            double J[4];
            jacobian(xi, J);
            return J[0] * J[3] - J[1] * J[2];
        }
        virtual double area() const = 0;
    };

    class QuadElement : public Element
    {
    public:
        QuadElement() : xc{{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}} {} // slot of a mesh's element store, assigned later

        /// X has shape (2, 4): the corners in counter-clockwise order
        explicit QuadElement(const double *X)
        {
            for (int c = 0; c < 4; ++c)
            {
                xc[c][0] = X[2 * c];
                xc[c][1] = X[2 * c + 1];
            }
        }

        void physical_coordinates(const double *xi, double *x) const override;
        void jacobian(const double *xi, double *J) const override;
        double area() const override;

        const double *corner(int i) const { return xc[i]; }

    private:
        double xc[4][2];
    };
} // namespace cuddh

#endif
