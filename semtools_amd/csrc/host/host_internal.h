// host_internal.h -- helpers shared by the host layer's translation units (host.cpp: strings, tokenizers, the search
// module; store.cpp: the workspace and its store; output.cpp: the reference's text / JSON output).
#pragma once
#include <algorithm>
#include <exception>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "host.h"

namespace semtools {

// throws Error(what + ": " + smt_last_error()) unless rc == SMT_OK
void check(int rc, const char *what);
void write_file_atomic(const std::string &path, const std::string &data);   // sibling + rename
bool path_exists(const std::string &p);
void mkdir_p(const std::string &dir);

// fn(begin, end) over [0, n) in contiguous slices on up to 8 host threads (one slice per `grain` items at least); an exception of
// any slice reaches the caller -- never std::terminate from a worker.
template <typename Fn>
inline void parallel_slices(size_t n, size_t grain, Fn fn)
{
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)8, n / std::max<size_t>(grain, 1)}));
    if (n_threads == 1) { fn((size_t)0, n); return; }
    std::vector<std::exception_ptr> failed(n_threads);
    std::vector<std::thread> th;
    auto run = [&](size_t t) {
        try { fn(n * t / n_threads, n * (t + 1) / n_threads); } catch (...) { failed[t] = std::current_exception(); }
    };
    // the last slice runs here; so does every slice whose thread could not be started
    size_t started = 0;
    try {
        for (; started + 1 < n_threads; ++started) th.emplace_back(run, started);
    } catch (const std::system_error &) {}
    for (size_t t = started; t < n_threads; ++t) run(t);
    for (auto &x : th) x.join();
    for (auto &f : failed) if (f) std::rethrow_exception(f);
}

}  // namespace semtools
