// Abstract linear operators acting on DEVICE vectors.
// Contract: reference include/Operator.hpp:6-26.
#ifndef CUDDH_AMD_OPERATOR_HPP
#define CUDDH_AMD_OPERATOR_HPP

namespace cuddh
{
    /// double precision operator
    class Operator
    {
    public:
        Operator() = default;
        virtual ~Operator() = default;

        /// y <- y + c * A * x
        virtual void action(double c, const double *x, double *y) const = 0;
        /// y <- A * x
        virtual void action(const double *x, double *y) const = 0;
    };

    /// single precision operator (the DDH interface operator works on float traces)
    class SinglePrecisionOperator
    {
    public:
        SinglePrecisionOperator() = default;
        virtual ~SinglePrecisionOperator() = default;

        /// y <- A * x
        virtual void action(const float *x, float *y) const = 0;
    };
} // namespace cuddh

#endif
