// host_internal.h -- helpers shared by the host layer's translation units (host.cpp: strings, tokenizers, the search
// module; store.cpp: the workspace and its store; output.cpp: the reference's text / JSON output).
#pragma once
#include <string>

#include "host.h"

namespace semtools {

// throws Error(what + ": " + smt_last_error()) unless rc == SMT_OK
void check(int rc, const char *what);
void write_file_atomic(const std::string &path, const std::string &data);   // sibling + rename
bool path_exists(const std::string &p);
void mkdir_p(const std::string &dir);

}  // namespace semtools
