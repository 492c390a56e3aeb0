// The two abstract operator interfaces every solver in this library is written against.  All vectors are DEVICE
// pointers.  Contract: reference include/Operator.hpp:6-26 (same class names and virtual signatures, so user
// operators written for the reference -- e.g. the Poisson and Helmholtz classes of its examples -- derive unchanged).
#ifndef CUDDH_AMD_OPERATOR_HPP
#define CUDDH_AMD_OPERATOR_HPP

namespace cuddh
{
    /// fp64 operator; matrix-free operators, preconditioners and user operators derive from it
    class Operator
    {
    protected:
        Operator() {}

    public:
        virtual ~Operator() {}
        virtual void action(double scale, const double *in, double *accum) const = 0; ///< accum += scale * A in
        virtual void action(const double *in, double *out) const = 0;                 ///< out = A in (overwrites)
    };

    /// Marker (not in the reference): an operator whose action() only queues work on cuddh::stream() -- no host-visible side effect,
    /// no synchronisation the caller could observe.  gmres() may then queue the NEXT Arnoldi step's action() before it has looked at
    /// the current step's Hessenberg column (one product may be computed and discarded when the iteration stops).  The library's
    /// own device operators carry it; user operators (callbacks, operators that count their calls) do not and are driven strictly
    /// in the reference's order.
    struct QueuesDeviceWorkOnly
    {
        virtual ~QueuesDeviceWorkOnly() = default;
    };

    /// fp32 operator: DDH acts on float trace vectors (reference include/DDH.hpp:22)
    class SinglePrecisionOperator
    {
    protected:
        SinglePrecisionOperator() {}

    public:
        virtual ~SinglePrecisionOperator() {}
        virtual void action(const float *in, float *out) const = 0; ///< out = A in
    };
} // namespace cuddh

#endif
