// Host-side parallel loops for the set-up phase (mesh, numbering, DDH tables): plain std::thread workers over
// contiguous index ranges.  The hot path never runs on the host; this only takes the one-off constructors of the
// reference's API (source/Mesh2D.cpp, H1Space.cpp, EnsembleSpace.cpp, DDH.cpp:323-609 are single-threaded host loops) off
// the critical path at 1024^2 elements.  Every use is written so that the result does not depend on the thread count.
#ifndef CUDDH_AMD_PARALLEL_HPP
#define CUDDH_AMD_PARALLEL_HPP

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <thread>
#include <vector>

namespace cuddh
{
    namespace detail
    {
        /// worker threads for set-up loops: CUDDH_SETUP_THREADS, else min(hardware threads, 16)
        inline int setup_threads()
        {
            static const int n = []
            {
                if (const char *e = std::getenv("CUDDH_SETUP_THREADS"))
                    return std::max(1, std::atoi(e));
                const unsigned hw = std::thread::hardware_concurrency();
                return static_cast<int>(std::min(16u, std::max(1u, hw)));
            }();
            return n;
        }

        /// how many contiguous chunks parallel_for will cut [0, n) into (one per worker)
        inline int chunk_count(std::size_t n, std::size_t min_chunk = 2048)
        {
            const std::size_t want = std::max<std::size_t>(1, n / std::max<std::size_t>(1, min_chunk));
            return static_cast<int>(std::min<std::size_t>(static_cast<std::size_t>(setup_threads()), want));
        }

        /// fn(begin, end, chunk) on contiguous chunks of [0, n); chunk c covers [n*c/C, n*(c+1)/C)
        template <typename F>
        void parallel_for(std::size_t n, F &&fn, std::size_t min_chunk = 2048)
        {
            const int C = chunk_count(n, min_chunk);
            auto range = [&](int c) { return std::pair<std::size_t, std::size_t>(n * c / C, n * (c + 1) / C); };
            if (C <= 1)
            {
                if (n > 0)
                    fn(std::size_t(0), n, 0);
                return;
            }
            std::vector<std::thread> workers;
            workers.reserve(C - 1);
            for (int c = 1; c < C; ++c)
                workers.emplace_back([&, c] { const auto r = range(c); fn(r.first, r.second, c); });
            const auto r0 = range(0);
            fn(r0.first, r0.second, 0);
            for (auto &w : workers)
                w.join();
        }
        /// CUDDH_SETUP_TIMING=1 prints the wall time of each constructor phase (host work, once per solver)
        struct PhaseTimer
        {
            const bool on = std::getenv("CUDDH_SETUP_TIMING") != nullptr;
            std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
            void lap(const char *what)
            {
                if (!on)
                    return;
                const auto t1 = std::chrono::steady_clock::now();
                std::fprintf(stderr, "[cuddh setup] %-36s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
                t0 = t1;
            }
        };
    } // namespace detail
} // namespace cuddh

#endif
