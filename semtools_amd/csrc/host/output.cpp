// output.cpp -- the reference's stdout formats, byte for byte: print_search_results / print_workspace_search_results
// (src/cmds/search.rs:35-110) and the JSON forms (src/json_mode.rs, src/cmds/search.rs:23-32, 208-241).
#include "host.h"
#include "host_internal.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "fmt.h"
#include "json.h"

namespace semtools {

// ================================================================== output
namespace cmds {

static void push_line(std::string &out, size_t line_number_1based, const std::string &line, bool highlight)
{
    char num[32];
    snprintf(num, sizeof(num), "%4zu: ", line_number_1based);  // "{:4}: {}"
    if (highlight) out += "\x1b[43m\x1b[30m";
    out += num;
    out += line;
    if (highlight) out += "\x1b[0m";
    out += "\n";
}

std::string print_search_results(const std::vector<search::SearchResult> &results, bool is_tty)
{
    std::string out;
    for (auto &r : results) {
        out += r.filename + ":" + std::to_string(r.start) + "::" + std::to_string(r.end) + " (" +
               fmt::rust_display(r.distance) + ")\n";                                   // search.rs:43
        for (size_t i = 0; i < r.lines.size(); ++i) {
            const size_t line_number = r.start + i;
            push_line(out, line_number + 1, r.lines[i], is_tty && line_number == r.match_line);  // :47-59
        }
        out += "\n";                                                                     // :61
    }
    return out;
}

std::string print_workspace_search_results(const std::vector<workspace::RankedLine> &ranked, size_t n_lines, bool is_tty)
{
    std::string out;
    for (auto &rl : ranked) {
        const size_t match = (size_t)rl.line_number;
        const size_t start = match > n_lines ? match - n_lines : 0;
        const size_t end = match + n_lines + 1;  // NOT clamped in the header (search.rs:77-79)
        out += rl.path + ":" + std::to_string(start) + "::" + std::to_string(end) + " (" + fmt::rust_display(rl.distance) + ")\n";
        std::string content;
        bool ok = true;
        try { content = read_to_string(rl.path); } catch (const Error &) { ok = false; }
        if (ok) {
            const std::vector<std::string> lines = lines_of(content);
            const size_t actual_end = std::min(end, lines.size());
            // the reference slices lines[start..actual_end] and panics if start > len (stale rows);
            // we print nothing for that window instead ("do not replicate")
            for (size_t ln = start; ln < actual_end; ++ln) push_line(out, ln + 1, lines[ln], is_tty && ln == match);
        } else {
            out += "    [Error: Could not read file content]\n";
        }
        out += "\n";
    }
    return out;
}

static json::Value result_json(const std::string &filename, size_t start, size_t end, size_t match, double distance,
                               const std::string &content)
{
    json::Value o = json::Value::object();  // field order: src/json_mode.rs:17-30
    o.set("filename", json::Value::str(filename));
    o.set("start_line_number", json::Value::uint(start));
    o.set("end_line_number", json::Value::uint(end));
    o.set("match_line_number", json::Value::uint(match));
    o.set("distance", json::Value::num(distance));
    o.set("content", json::Value::str(content));
    return o;
}

static std::string join_lines(const std::vector<std::string> &lines, size_t b, size_t e)
{
    std::string s;
    for (size_t i = b; i < e; ++i) { if (i > b) s += "\n"; s += lines[i]; }
    return s;
}

std::string search_results_json(const std::vector<search::SearchResult> &results)
{
    json::Value arr = json::Value::array();
    for (auto &r : results)
        arr.arr.push_back(result_json(r.filename, r.start, r.end, r.match_line, r.distance, join_lines(r.lines, 0, r.lines.size())));
    json::Value root = json::Value::object();
    root.set("results", std::move(arr));
    return json::to_string_pretty(root) + "\n";
}

std::string workspace_results_json(const std::vector<workspace::RankedLine> &ranked, size_t n_lines)
{
    json::Value arr = json::Value::array();
    for (auto &rl : ranked) {
        const size_t match = (size_t)rl.line_number;
        const size_t start = match > n_lines ? match - n_lines : 0;
        const size_t end = match + n_lines + 1;
        std::string content;
        try {
            const std::vector<std::string> lines = lines_of(read_to_string(rl.path));
            const size_t actual_end = std::min(end, lines.size());
            content = start <= actual_end ? join_lines(lines, start, actual_end) : "";
        } catch (const Error &) {
            content = "[Error: Could not read file content]";
        }
        arr.arr.push_back(result_json(rl.path, start, end, match, (double)rl.distance, content));  // `as f64` (search.rs:233)
    }
    json::Value root = json::Value::object();
    root.set("results", std::move(arr));
    return json::to_string_pretty(root) + "\n";
}

}  // namespace cmds
}  // namespace semtools
