// cli.cpp -- `semtools` CLI replica for the search / workspace subcommands
// (reference: src/bin/semtools.rs:52-83,122-131,134-206; src/cmds/search.rs:113-276;
// src/cmds/workspace.rs).  Same flags, same stdout/stderr split, same output bytes.
// The model comes from $SEMTOOLS_MODEL_DIR (model.safetensors + vocab.txt [+ config.json]):
// the HF-hub download of StaticModel::from_pretrained is out of scope (no network here).
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/semtools_host.h"

static int die(const std::string &msg)
{
    fprintf(stderr, "Error: %s\n", msg.c_str());
    return 1;
}

// `--max-distance NaN`: the reference keeps a row iff distance < max_distance (src/search/mod.rs:88-89), which is
// false for every row when max_distance is NaN -- it prints nothing.  The C ABI uses NaN for "no threshold", so the
// CLI maps a NaN argument to a threshold nothing passes (-1: cosine distances are >= 0).
static double parse_max_distance(const char *text)
{
    const double v = strtod(text, nullptr);
    return v != v ? -1.0 : v;
}

static int usage()
{
    fprintf(stderr,
            "Usage: semtools <COMMAND>\n\nCommands:\n"
            "  search     A CLI tool for fast semantic keyword search\n"
            "  workspace  Manage semtools workspaces\n\n"
            "semtools search <QUERY> [FILES]... [-n, --n-lines <N>] [--top-k <K>] [-m, --max-distance <D>]\n"
            "                [-i, --ignore-case] [-j, --json] [-w, --workspace <NAME>]\n"
            "semtools workspace [-j, --json] use <NAME> | status [NAME] | prune [NAME] | reembed [NAME]\n"
            "               (reembed is not in the reference: re-creates the stored vectors from cached token ids with\n"
            "                the model in $SEMTOOLS_MODEL_DIR -- a new table, same tokenizer -- without reading the files)\n"
            "semtools serve <FILES>... [-n N] [--top-k K] [-m D] [-i] [-j] [--batch B]\n"
            "               (resident mode, not in the reference: files embedded once, one query per stdin line;\n"
            "                each answer is what `semtools search <query> <FILES>` prints, preceded by `### <query>`)\n");
    return 2;
}

int main(int argc, char **argv)
{
    if (argc < 2) return usage();
    const std::string cmd = argv[1];
    std::vector<std::string> args(argv + 2, argv + argc);

    // The GPUs this process owns.  $SEMTOOLS_DEVICES = "0,1,2,3" | "all" | "<device>:<logical shards>" runs the whole
    // path sharded over them (table replicated, lines and rows dealt over the GPUs, one all-gather per search); unset, it is
    // one GPU ($SEMTOOLS_DEVICE, default 0).  Output bytes do not depend on it.
    smt_group *group = nullptr;
    auto need_ctx = [&]() -> bool {
        if (group) return true;
        const char *devs = getenv("SEMTOOLS_DEVICES"), *dev = getenv("SEMTOOLS_DEVICE");
        const std::string spec = (devs && *devs) ? devs : (dev && *dev) ? dev : "0";
        if (smt_host_group_from_spec(spec.c_str(), &group) != SMT_OK) { die(smt_last_error()); return false; }
        return true;
    };

    if (cmd == "workspace") {
        bool json = false;
        std::vector<std::string> pos;
        for (auto &a : args) { if (a == "-j" || a == "--json") json = true; else pos.push_back(a); }
        if (pos.empty()) return usage();
        char *text = nullptr;
        int rc;
        if (pos[0] == "use") {
            if (pos.size() < 2) return usage();
            if (json && !need_ctx()) return 1;
            rc = smt_host_workspace_use_group(group, pos[1].c_str(), json, &text);
        } else if (pos[0] == "reembed") {
            // not in the reference: re-create the stored vectors from the cached token ids with the model in
            // $SEMTOOLS_MODEL_DIR (a new embedding table behind the same tokenizer), no source file is read
            if (!need_ctx()) return 1;
            const char *model_dir = getenv("SEMTOOLS_MODEL_DIR");
            if (!model_dir) return die("SEMTOOLS_MODEL_DIR is not set (directory with model.safetensors + vocab.txt)");
            smt_host_model *model = nullptr;
            if (smt_host_model_from_dir_group(group, model_dir, &model) != SMT_OK) return die(smt_last_error());
            rc = smt_host_workspace_reembed(model, pos.size() > 1 ? pos[1].c_str() : nullptr, json, &text);
            smt_host_model_destroy(model);
        } else if (pos[0] == "status" || pos[0] == "prune") {
            if (!need_ctx()) return 1;
            const char *nm = pos.size() > 1 ? pos[1].c_str() : nullptr;
            rc = pos[0] == "status" ? smt_host_workspace_status_group(group, nm, json, &text) : smt_host_workspace_prune_group(group, nm, json, &text);
        } else return usage();
        if (rc != SMT_OK) return die(smt_last_error());
        fputs(text, stdout);
        smt_host_free(text);
        smt_group_destroy(group);
        return 0;
    }
    if (cmd == "serve") {
        std::vector<std::string> files;
        uint64_t n_lines = 3, top_k = 3, batch = 64;
        double max_distance = NAN;
        bool ignore_case = false, json = false;
        for (size_t i = 0; i < args.size(); ++i) {
            const std::string &a = args[i];
            auto val = [&](const char *name) -> const char * {
                if (i + 1 >= args.size()) { fprintf(stderr, "error: a value is required for '%s'\n", name); exit(2); }
                return args[++i].c_str();
            };
            if (a == "-n" || a == "--n-lines" || a == "--context") n_lines = strtoull(val("--n-lines"), nullptr, 10);
            else if (a == "--top-k") top_k = strtoull(val("--top-k"), nullptr, 10);
            else if (a == "-m" || a == "--max-distance" || a == "--threshold") max_distance = parse_max_distance(val("--max-distance"));
            else if (a == "--batch") batch = std::max<uint64_t>(1, strtoull(val("--batch"), nullptr, 10));
            else if (a == "-i" || a == "--ignore-case") ignore_case = true;
            else if (a == "-j" || a == "--json") json = true;
            else files.push_back(a);
        }
        if (files.empty()) return usage();
        if (!need_ctx()) return 1;
        const char *model_dir = getenv("SEMTOOLS_MODEL_DIR");
        if (!model_dir) return die("SEMTOOLS_MODEL_DIR is not set (directory with model.safetensors + vocab.txt)");
        smt_host_model *model = nullptr;
        if (smt_host_model_from_dir_group(group, model_dir, &model) != SMT_OK) return die(smt_last_error());
        std::vector<const char *> fp;
        for (auto &f : files) fp.push_back(f.c_str());
        smt_host_session *session = nullptr;
        if (smt_host_session_open(model, fp.data(), fp.size(), ignore_case, &session) != SMT_OK) return die(smt_last_error());
        fprintf(stderr, "semtools serve: %llu lines resident\n", (unsigned long long)smt_host_session_lines(session));
        const int is_tty = isatty(STDOUT_FILENO);
        std::vector<std::string> pending;
        std::string line;
        auto flush_batch = [&]() -> int {
            if (pending.empty()) return 0;
            std::vector<const char *> qp;
            for (auto &q : pending) qp.push_back(q.c_str());
            std::vector<char *> outs(pending.size(), nullptr);
            if (smt_host_session_search(session, qp.data(), qp.size(), n_lines, top_k, max_distance, json, is_tty, outs.data()) != SMT_OK)
                return die(smt_last_error());
            for (size_t i = 0; i < pending.size(); ++i) {
                printf("### %s\n", pending[i].c_str());
                fputs(outs[i], stdout);
                smt_host_free(outs[i]);
            }
            fflush(stdout);
            pending.clear();
            return 0;
        };
        while (std::getline(std::cin, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty()) { if (flush_batch()) return 1; continue; }  // blank line = answer what is queued
            pending.push_back(line);
            if (pending.size() >= batch && flush_batch()) return 1;
        }
        if (flush_batch()) return 1;
        if (const char *t = getenv("SEMTOOLS_TIMING"); t && *t == '1') {
            char *phases = smt_host_timing_json();
            fprintf(stderr, "{\"timing_ms\": %s}\n", phases ? phases : "{}");
            smt_host_free(phases);
        }
        smt_host_session_close(session);
        smt_host_model_destroy(model);
        smt_group_destroy(group);
        return 0;
    }
    if (cmd != "search") return usage();

    std::string query;
    std::vector<std::string> files;
    uint64_t n_lines = 3, top_k = 3;
    double max_distance = NAN;
    bool ignore_case = false, json = false, have_query = false;
    const char *workspace = nullptr;
    std::string ws_store;
    for (size_t i = 0; i < args.size(); ++i) {
        const std::string &a = args[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= args.size()) { fprintf(stderr, "error: a value is required for '%s'\n", name); exit(2); }
            return args[++i].c_str();
        };
        if (a == "-n" || a == "--n-lines" || a == "--context") n_lines = strtoull(val("--n-lines"), nullptr, 10);
        else if (a == "--top-k") top_k = strtoull(val("--top-k"), nullptr, 10);
        else if (a == "-m" || a == "--max-distance" || a == "--threshold") max_distance = parse_max_distance(val("--max-distance"));
        else if (a == "-i" || a == "--ignore-case") ignore_case = true;
        else if (a == "-j" || a == "--json") json = true;
        else if (a == "-w" || a == "--workspace") { ws_store = val("--workspace"); workspace = ws_store.c_str(); }
        else if (!have_query) { query = a; have_query = true; }
        else files.push_back(a);
    }
    if (!have_query) return usage();

    if (!need_ctx()) return 1;
    const char *model_dir = getenv("SEMTOOLS_MODEL_DIR");
    if (!model_dir) return die("SEMTOOLS_MODEL_DIR is not set (directory with model.safetensors + vocab.txt)");
    smt_host_model *model = nullptr;
    if (smt_host_model_from_dir_group(group, model_dir, &model) != SMT_OK) return die(smt_last_error());

    const int is_tty = isatty(STDOUT_FILENO);
    char *text = nullptr;
    int rc = SMT_OK;
    bool done = false;

    // stdin input (non-workspace mode): src/cmds/search.rs:145-176
    if (files.empty() && !isatty(STDIN_FILENO)) {
        std::stringstream ss;
        ss << std::cin.rdbuf();
        const std::string content = ss.str();
        if (!content.empty()) {
            rc = smt_host_search_content(model, query.c_str(), "<stdin>", content.c_str(), n_lines, top_k, max_distance,
                                         ignore_case, json, is_tty, &text);
            done = true;
        }
    }
    if (!done && files.empty()) {  // src/cmds/search.rs:178-193
        const char *msg = "No input provided. Either specify files as arguments or pipe input to stdin.";
        if (json) fprintf(stderr, "{\n  \"error\": \"%s\",\n  \"error_type\": \"NoInput\"\n}\n", msg);
        else fprintf(stderr, "Error: %s\n", msg);
        return 1;
    }
    if (!done) {
        std::vector<const char *> fp;
        for (auto &f : files) fp.push_back(f.c_str());
        const char *env_ws = getenv("SEMTOOLS_WORKSPACE");
        const bool ws_active = workspace != nullptr || (env_ws && *env_ws);  // Workspace::active(..).is_ok()
        if (ws_active)
            rc = smt_host_search_workspace(model, query.c_str(), fp.data(), fp.size(), n_lines, top_k, max_distance, ignore_case,
                                           workspace, json, is_tty, &text);
        else
            rc = smt_host_search_files(model, query.c_str(), fp.data(), fp.size(), n_lines, top_k, max_distance, ignore_case, json,
                                       is_tty, &text);
    }
    if (rc != SMT_OK) return die(smt_last_error());
    fputs(text, stdout);
    fflush(stdout);
    smt_host_free(text);
    if (const char *t = getenv("SEMTOOLS_TIMING"); t && *t == '1') {  // where the wall time of this invocation went
        char *phases = smt_host_timing_json();
        fprintf(stderr, "{\"timing_ms\": %s}\n", phases ? phases : "{}");
        smt_host_free(phases);
    }
    smt_host_model_destroy(model);
    smt_group_destroy(group);
    return 0;
}
