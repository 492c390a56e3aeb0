// Raw native-endian fp64 dump used by the example drivers
// (same file format as reference examples/examples.hpp:11-16).
#ifndef CUDDH_AMD_EXAMPLES_HPP
#define CUDDH_AMD_EXAMPLES_HPP

#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>

namespace cuddh
{
    inline void to_file(const std::string &fname, int n_dof, const double *u)
    {
        std::ofstream out(fname, std::ios::binary);
        out.write(reinterpret_cast<const char *>(u), static_cast<std::streamsize>(n_dof) * sizeof(double));
    }
} // namespace cuddh

#endif
