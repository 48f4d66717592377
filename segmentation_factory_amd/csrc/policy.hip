// The dispatch policy (policy.h) and the launch trace: host code only.
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/segfac.h"
#include "policy.h"

namespace {

struct Row { const char* field; const char* env; int def; const char* doc; int SegfPolicy::*member; };
#define SEGF_POLICY_ROW(field, env, def, doc) {#field, env, def, doc, &SegfPolicy::field},
const Row kRows[] = {SEGF_POLICY_TABLE(SEGF_POLICY_ROW)};
#undef SEGF_POLICY_ROW
constexpr int kCount = (int)(sizeof(kRows) / sizeof(kRows[0]));

SegfPolicy g_policy;
std::once_flag g_once;

int parse(const char* env, int def) {
    const char* e = getenv(env);
    if (!e) return def;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e) return 1;          // set, but not a number ("", "yes"): on
    return (int)v;
}
void load() {
    for (int i = 0; i < kCount; ++i) g_policy.*(kRows[i].member) = parse(kRows[i].env, kRows[i].def);
}
const Row* find(const char* name) {
    if (!name) return nullptr;
    for (int i = 0; i < kCount; ++i)
        if (!strcmp(name, kRows[i].field) || !strcmp(name, kRows[i].env)) return &kRows[i];
    return nullptr;
}
void ensure() {
    std::call_once(g_once, [] { g_policy.g8_stagger_fp8 = 1; load(); });
}

thread_local SegfTrace t_trace = {0, 0, 0, {nullptr}, {nullptr}};

}   // namespace

const SegfPolicy& segf_policy() {
    ensure();
    return g_policy;
}

SegfTrace& segf_trace() { return t_trace; }
void segf_trace_note(const char* kernel_text, const char* where) {
    SegfTrace& t = t_trace;
    if (t.n < (int)(sizeof(t.names) / sizeof(t.names[0]))) { t.names[t.n] = kernel_text; t.where[t.n] = where; }
    ++t.n;
}

extern "C" int segf_policy_count(void) { return kCount; }
extern "C" int segf_policy_describe(int i, const char** field, const char** env, int* value, int* def, const char** doc) {
    if (i < 0 || i >= kCount) return SEGF_ERR_SHAPE;
    ensure();
    if (field) *field = kRows[i].field;
    if (env) *env = kRows[i].env;
    if (value) *value = g_policy.*(kRows[i].member);
    if (def) *def = kRows[i].def;
    if (doc) *doc = kRows[i].doc;
    return 0;
}
extern "C" int segf_policy_get(const char* name) {
    const Row* r = find(name);
    if (!r) return INT_MIN;
    ensure();
    return g_policy.*(r->member);
}
extern "C" int segf_policy_set(const char* name, int value) {
    const Row* r = find(name);
    if (!r) return INT_MIN;
    ensure();
    const int prev = g_policy.*(r->member);
    g_policy.*(r->member) = value;
    return prev;
}
extern "C" void segf_policy_reload(void) {
    ensure();
    load();
}
extern "C" int segf_gemm8_option(int what, int value) {      // what 0: stagger of the fp8 eight-phase kernels (returns the previous value)
    if (what != 0) return SEGF_ERR_SHAPE;
    ensure();
    const int prev = g_policy.g8_stagger_fp8;
    if (value == 0 || value == 1) g_policy.g8_stagger_fp8 = value;
    return prev;
}

extern "C" void segf_trace_begin(int dry_run) {
    SegfTrace& t = t_trace;
    t.on = 1;
    t.dry = dry_run ? 1 : 0;
    t.n = 0;
}
// Writes the kernels launched on this thread since segf_trace_begin, one instantiated name per line, into buf (always terminated;
// truncated when cap is too small) and switches the trace (and the dry run) off.  Returns the number of launches.
extern "C" int segf_trace_end(char* buf, int cap) {
    SegfTrace& t = t_trace;
    const int n = t.n;
    if (buf && cap > 0) {
        int pos = 0;
        buf[0] = 0;
        const int kept = n < (int)(sizeof(t.names) / sizeof(t.names[0])) ? n : (int)(sizeof(t.names) / sizeof(t.names[0]));
        for (int i = 0; i < kept; ++i) {
            // "(gemm_skinny_kernel<L, KS, NT>)" + "bool gemm_skinny_launch(...) [L = 0]" -> "gemm_skinny_kernel<L, KS, NT> [L = 0]"
            const char* s = t.names[i];
            int len = (int)strlen(s);
            if (len >= 2 && s[0] == '(' && s[len - 1] == ')') { ++s; len -= 2; }
            const char* w = t.where[i] ? strstr(t.where[i], " [") : nullptr;
            const int wlen = w ? (int)strlen(w) : 0;
            if (pos + len + wlen + 2 > cap) break;
            memcpy(buf + pos, s, (size_t)len);
            pos += len;
            if (wlen) { memcpy(buf + pos, w, (size_t)wlen); pos += wlen; }
            buf[pos++] = '\n';
            buf[pos] = 0;
        }
    }
    t.on = t.dry = t.n = 0;
    return n;
}
