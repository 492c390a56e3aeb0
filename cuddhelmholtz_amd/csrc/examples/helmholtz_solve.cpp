// Native C++ driver for the global Helmholtz operator (BASELINE config 2 shape): unpreconditioned restarted GMRES on the
// fused complex Helmholtz apply, written against csrc/include/cuddh.hpp.
//   helmholtz_solve [nx=256] [n_basis=4] [omega_over_pi=8] [gmres_m=20] [maxit=10] [tol=1e-6] [out_dir=-] [ordering=native|reference]
// ordering native (default): HelmholtzOperator::gmres -- the iteration vectors live in the plan's own ordering (permuted once at
// entry and exit); reference: gmres() on [u; v] in H1Space numbering, the call the reference's examples make.
// a(x) = 0.2 inside the disk of radius 1/4, 1 elsewhere (interpolated at the nodes; a = 1 on the boundary), two Gaussian
// sources as in the reference's examples.  Prints one summary line; writes <out_dir>/xy.0000 and helmholtz.0000 unless "-".
#include <chrono>
#include <cstdlib>
#include <string>

#include "cuddh.hpp"
#include "cuddh_hip.h"
#include "examples.hpp"

using namespace cuddh;

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? std::atoi(argv[1]) : 256;
    const int nb = argc > 2 ? std::atoi(argv[2]) : 4;
    const double omega = M_PI * (argc > 3 ? std::atof(argv[3]) : 8.0);
    const int m = argc > 4 ? std::atoi(argv[4]) : 20;
    const int maxit = argc > 5 ? std::atoi(argv[5]) : 10;
    const double tol = argc > 6 ? std::atof(argv[6]) : 1e-6;
    const std::string out_dir = argc > 7 ? argv[7] : "-";
    const bool native = !(argc > 8 && std::string(argv[8]) == "reference");

    Mesh2D mesh = Mesh2D::uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0);
    Basis basis(nb);
    H1Space fem(mesh, basis);
    ivec boundary = mesh.boundary_edges();
    FaceSpace fs(fem, boundary.size(), boundary);
    const int ndof = fem.size(), N = 2 * ndof;

    host_device_dvec U(N), b(N), a2(ndof), ax(fs.size());
    {
        const double *xy = fem.physical_coordinates(MemorySpace::HOST);
        double *h = a2.host_write();
        for (int i = 0; i < ndof; ++i)
        {
            const double x = xy[2 * i], y = xy[2 * i + 1];
            h[i] = (x * x + y * y < 0.0625) ? 0.04 : 1.0;
        }
        double *hf = ax.host_write();
        for (int i = 0; i < fs.size(); ++i)
            hf[i] = 1.0;
    }
    double *d_U = U.device_write(), *d_b = b.device_write();
    LinearFunctional l(fem);
    l.action([=] __device__(const double X[2]) -> double
    {
        const double s = omega * omega;
        const double r0 = (X[0] + 0.5) * (X[0] + 0.5) + X[1] * X[1];
        const double r1 = (X[0] - 0.5) * (X[0] - 0.5) + (X[1] + 0.5) * (X[1] + 0.5);
        return s / M_PI * (exp(-s * r0) + exp(-s * r1));
    }, d_b);

    HelmholtzOperator A(omega, a2.device_read(), ax.device_read(), fem, fs);

    using clk = std::chrono::steady_clock;
    detail::check_hip(cuddh_hip_stream_sync(stream()), "sync");
    const auto t0 = clk::now();
    solver_out out = native ? A.gmres(d_U, d_b, m, maxit, tol, 0) : gmres(N, d_U, &A, d_b, m, maxit, tol, 0);
    detail::check_hip(cuddh_hip_stream_sync(stream()), "sync");
    const double t_gmres = std::chrono::duration<double>(clk::now() - t0).count();

    const double *h_U = U.host_read();
    if (out_dir != "-")
    {
        to_file(out_dir + "/xy.0000", N, fem.physical_coordinates(MemorySpace::HOST));
        to_file(out_dir + "/helmholtz.0000", N, h_U);
    }
    double unorm = 0.0;
    for (int i = 0; i < N; ++i)
        unorm += h_U[i] * h_U[i];
    std::cout << "helmholtz_solve nx=" << nx << " nb=" << nb << " omega/pi=" << omega / M_PI << " N=" << N << " fused=" << A.fused()
              << " ordering=" << ((native && A.has_native()) ? "native" : "reference")
              << " success=" << out.success << " num_iter=" << out.num_iter << " num_matvec=" << out.num_matvec
              << " rel_res=" << out.res_norm.back() / out.res_norm.front() << " |U|=" << std::sqrt(unorm) << " t_gmres=" << t_gmres
              << " DoF*iter/s=" << static_cast<double>(N) * out.num_matvec / t_gmres
              << " us_per_matvec=" << 1e6 * t_gmres / out.num_matvec << std::endl;
    return 0;
}
