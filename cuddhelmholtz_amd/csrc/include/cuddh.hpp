// Umbrella header: everything a driver written against the reference's cuddh.hpp
// needs (reference cuddh.hpp:6-26), so that examples/DDH.cpp and
// examples/Poisson.cpp compile unchanged with `hipcc -x hip`.
#ifndef CUDDH_HPP
#define CUDDH_HPP

#include "cuddh_config.hpp"

#include "cuddh/error.hpp"
#include "cuddh/tensor.hpp"
#include "cuddh/memory.hpp"
#include "cuddh/launch.hpp"
#include "cuddh/quadrature.hpp"
#include "cuddh/basis.hpp"
#include "cuddh/geometry.hpp"
#include "cuddh/mesh.hpp"
#include "cuddh/meshio.hpp"
#include "cuddh/operator.hpp"
#include "cuddh/spaces.hpp"
#include "cuddh/ensemble.hpp"
#include "cuddh/blas1.hpp"
#include "cuddh/operators.hpp"
#include "cuddh/functionals.hpp"
#include "cuddh/krylov.hpp"
#include "cuddh/ddh.hpp"
#include "cuddh/helmholtz.hpp"
#include "cuddh/multigpu.hpp"
#include "cuddh/partition.hpp"

#endif
