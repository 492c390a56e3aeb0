// Force-included (-include) when the reference's own tests/*.cpp are compiled unchanged against csrc/include/cuddh.hpp
// (cuddhelmholtz_amd/build.py::build_examples).  The reference's loader does `std::string dir = UNSTRUCTURED_SQUARE_MESH_DIR;`
// (tests/load_unstructured_square.cpp:13) and its build system defines the macro as a path literal
// (tests/CMakeLists.txt:8).  Here the macro expands to this function, so the binary finds the fixture wherever the checkout is:
// $CUDDH_MESH_DIR if set, else <root>/tests/golden/unstructured_square with <root> = two directories above the binary
// (build/examples/reference_tests_driver).
#ifndef CUDDH_AMD_REFERENCE_TESTS_ENV_HPP
#define CUDDH_AMD_REFERENCE_TESTS_ENV_HPP

#include <limits.h>
#include <unistd.h>

#include <cstdlib>
#include <string>

inline std::string cuddh_reference_mesh_dir()
{
    if (const char *e = std::getenv("CUDDH_MESH_DIR"))
        return e;
    char buf[PATH_MAX];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    std::string exe = n > 0 ? std::string(buf, static_cast<std::size_t>(n)) : std::string(".");
    for (int up = 0; up < 3; ++up) // strip <name>, examples, build
    {
        const std::size_t cut = exe.find_last_of('/');
        exe = cut == std::string::npos ? std::string(".") : exe.substr(0, cut);
    }
    return exe + "/tests/golden/unstructured_square";
}

#endif
