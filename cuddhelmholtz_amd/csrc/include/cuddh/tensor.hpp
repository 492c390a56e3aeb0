// Column-major N-d array views (first index fastest) usable on host and device,
// plus an owning host tensor.  API contract: reference include/Tensor.hpp
// (TensorWrapper :53-239, reshape :247-263, Tensor :268-354, aliases :356-432).
#ifndef CUDDH_AMD_TENSOR_HPP
#define CUDDH_AMD_TENSOR_HPP

#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

#include "error.hpp"

namespace cuddh
{
    namespace detail
    {
        // fills extents[0..] from a pack and returns the product
        template <typename... Extents>
        __host__ __device__ inline int store_extents(int *extents, Extents... e)
        {
            const int vals[] = {static_cast<int>(e)...};
            int total = 1;
            for (unsigned d = 0; d < sizeof...(e); ++d)
            {
                if (vals[d] < 0)
                    cuddh_error("Tensor error: tensor cannot have negative dimensions.");
                extents[d] = vals[d];
                total *= vals[d];
            }
            return total;
        }

        // Horner evaluation of i0 + n0*(i1 + n1*(i2 + ...))
        template <typename... Indices>
        __host__ __device__ inline int flat_index(const int *extents, Indices... ids)
        {
            const int vals[] = {static_cast<int>(ids)...};
            constexpr int N = sizeof...(ids);
            int off = 0;
#pragma unroll
            for (int d = N - 1; d >= 0; --d)
            {
#ifdef CUDDH_DEBUG
                if (vals[d] < 0 || vals[d] >= extents[d])
                    cuddh_error("Tensor error: tensor index out of range.");
#endif
                off = vals[d] + extents[d] * off;
            }
            return off;
        }
    } // namespace detail

    /// Non-owning view of `Dim`-dimensional column-major data.
    template <int Dim, typename scalar>
    class TensorWrapper
    {
    public:
        __host__ __device__ TensorWrapper() : _shape{}, len(0), ptr(nullptr) {}

        template <typename... Sizes>
        __host__ __device__ explicit TensorWrapper(scalar *data_, Sizes... shape_) : ptr(data_)
        {
            static_assert(Dim > 0, "Tensor must have a positive number of dimensions");
            static_assert(sizeof...(shape_) == Dim, "Wrong number of dimensions specified.");
            len = detail::store_extents(_shape, shape_...);
        }

        template <typename... Indices>
        __host__ __device__ scalar &at(Indices... ids)
        {
            static_assert(sizeof...(ids) == Dim, "Wrong number of indices specified.");
#ifdef CUDDH_DEBUG
            if (ptr == nullptr)
                cuddh_error("TensorWrapper::at error: memory uninitialized.");
#endif
            return ptr[detail::flat_index(_shape, ids...)];
        }

        template <typename... Indices>
        __host__ __device__ const scalar &at(Indices... ids) const
        {
            static_assert(sizeof...(ids) == Dim, "Wrong number of indices specified.");
#ifdef CUDDH_DEBUG
            if (ptr == nullptr)
                cuddh_error("TensorWrapper::at error: memory uninitialized.");
#endif
            return ptr[detail::flat_index(_shape, ids...)];
        }

        template <typename... Indices>
        __host__ __device__ scalar &operator()(Indices... ids) { return at(ids...); }

        template <typename... Indices>
        __host__ __device__ const scalar &operator()(Indices... ids) const { return at(ids...); }

        __host__ __device__ scalar &operator[](int idx)
        {
#ifdef CUDDH_DEBUG
            if (ptr == nullptr)
                cuddh_error("TensorWrapper::operator[] error: memory uninitialized.");
            if (idx < 0 || idx >= len)
                cuddh_error("TensorWrapper::operator[] error: linear index out of range.");
#endif
            return ptr[idx];
        }

        __host__ __device__ const scalar &operator[](int idx) const
        {
#ifdef CUDDH_DEBUG
            if (ptr == nullptr)
                cuddh_error("TensorWrapper::operator[] error: memory uninitialized.");
            if (idx < 0 || idx >= len)
                cuddh_error("TensorWrapper::operator[] error: linear index out of range.");
#endif
            return ptr[idx];
        }

        /// implicit decay to the wrapped pointer (the examples rely on it)
        __host__ __device__ operator scalar *() { return ptr; }
        __host__ __device__ operator const scalar *() const { return ptr; }

        __host__ __device__ scalar *data() { return ptr; }
        __host__ __device__ const scalar *data() const { return ptr; }

        __host__ __device__ scalar *begin() { return ptr; }
        __host__ __device__ scalar *end() { return ptr + len; }
        __host__ __device__ const scalar *begin() const { return ptr; }
        __host__ __device__ const scalar *end() const { return ptr + len; }

        __host__ __device__ const int *shape() const { return _shape; }

        __host__ __device__ int shape(int d) const
        {
#ifdef CUDDH_DEBUG
            if (d < 0 || d >= Dim)
                cuddh_error("TensorWrapper::shape() error: shape index out of range of Dim.");
#endif
            return _shape[d];
        }

        __host__ __device__ int size() const { return len; }

    protected:
        int _shape[Dim];
        int len;
        scalar *ptr;
    };

    template <typename scalar, typename... Sizes>
    __host__ __device__ inline TensorWrapper<sizeof...(Sizes), scalar> reshape(scalar *data, Sizes... shape)
    {
        return TensorWrapper<sizeof...(Sizes), scalar>(data, shape...);
    }

    template <typename scalar, int Dim, typename... Sizes>
    __host__ __device__ inline TensorWrapper<sizeof...(Sizes), scalar> reshape(TensorWrapper<Dim, scalar> tensor, Sizes... shape)
    {
        return TensorWrapper<sizeof...(Sizes), scalar>(tensor.data(), shape...);
    }

    /// Host tensor that owns (zero-initialised) storage.
    template <int Dim, typename scalar>
    class Tensor : public TensorWrapper<Dim, scalar>
    {
        using view = TensorWrapper<Dim, scalar>;

    public:
        Tensor() : view() {}

        template <typename... Sizes, typename = std::enable_if_t<(std::is_integral_v<Sizes> && ...)>>
        explicit Tensor(Sizes... shape_) : view(nullptr, shape_...), store(this->len > 0 ? new scalar[this->len]() : nullptr)
        {
            this->ptr = store.get();
            capacity = this->len;
        }

        Tensor(const Tensor &other) : view() { assign(other); }

        Tensor &operator=(const Tensor &other)
        {
            if (this != &other)
                assign(other);
            return *this;
        }

        Tensor(Tensor &&other) noexcept : view(), store(std::move(other.store))
        {
            take_meta(other);
        }

        Tensor &operator=(Tensor &&other) noexcept
        {
            store = std::move(other.store);
            take_meta(other);
            return *this;
        }

        /// change the shape; storage is reallocated (and zeroed) only when it must grow
        template <typename... Sizes>
        void reshape(Sizes... shape_)
        {
            static_assert(sizeof...(shape_) == Dim, "Wrong number of dimensions specified.");
            int extents[Dim];
            const int n = detail::store_extents(extents, shape_...);
            if (n > capacity)
            {
                store.reset(new scalar[n]());
                capacity = n;
                this->ptr = store.get();
            }
            for (int d = 0; d < Dim; ++d)
                this->_shape[d] = extents[d];
            this->len = n;
        }

    private:
        void assign(const Tensor &other)
        {
            if (other.len > capacity || !store)
            {
                store.reset(other.len > 0 ? new scalar[other.len] : nullptr);
                capacity = other.len;
            }
            this->ptr = store.get();
            this->len = other.len;
            for (int d = 0; d < Dim; ++d)
                this->_shape[d] = other._shape[d];
            for (int i = 0; i < other.len; ++i)
                store[i] = other.ptr[i];
        }

        void take_meta(Tensor &other)
        {
            this->ptr = store.get();
            this->len = other.len;
            capacity = other.capacity;
            for (int d = 0; d < Dim; ++d)
                this->_shape[d] = other._shape[d];
            other.ptr = nullptr;
            other.len = 0;
            other.capacity = 0;
        }

        std::unique_ptr<scalar[]> store;
        int capacity = 0;
    };

    template <typename scalar> using VectorWrapper = TensorWrapper<1, scalar>;
    template <typename scalar> using MatrixWrapper = TensorWrapper<2, scalar>;
    template <typename scalar> using CubeWrapper = TensorWrapper<3, scalar>;

    typedef TensorWrapper<1, double> dvec_wrapper;
    typedef TensorWrapper<1, const double> const_dvec_wrapper;
    typedef TensorWrapper<2, double> dmat_wrapper;
    typedef TensorWrapper<2, const double> const_dmat_wrapper;
    typedef TensorWrapper<3, double> dcube_wrapper;
    typedef TensorWrapper<3, const double> const_dcube_wrapper;
    typedef TensorWrapper<1, int> ivec_wrapper;
    typedef TensorWrapper<1, const int> const_ivec_wrapper;
    typedef TensorWrapper<2, int> imat_wrapper;
    typedef TensorWrapper<2, const int> const_imat_wrapper;
    typedef TensorWrapper<3, int> icube_wrapper;
    typedef TensorWrapper<3, const int> const_icube_wrapper;

    template <typename scalar> using Vec = Tensor<1, scalar>;
    template <typename scalar> using Matrix = Tensor<2, scalar>;
    template <typename scalar> using Cube = Tensor<3, scalar>;

    typedef Vec<double> dvec;
    typedef Matrix<double> dmat;
    typedef Cube<double> dcube;
    typedef Vec<int> ivec;
    typedef Matrix<int> imat;
    typedef Cube<int> icube;
} // namespace cuddh

#endif
