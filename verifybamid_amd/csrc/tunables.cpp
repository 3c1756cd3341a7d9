// tunables.cpp -- see tunables.h.
#include "tunables.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace vb2 {

namespace {

struct Entry {
    const char* name;
    int Tunables::*field;
};
const Entry kEntries[] = {
#define VB2_X(name, dflt) {#name, &Tunables::name},
    VB2_TUNABLE_LIST(VB2_X)
#undef VB2_X
};

// "3" -> 3; the words the scripts of earlier rounds used (VB2_REDUCE=ticket|tagged) keep working; any other
// non-empty text counts as 1
int parse_value(const char* v)
{
    char* end = nullptr;
    const long n = std::strtol(v, &end, 10);
    if (end != v && *end == '\0') return (int)n;
    if (!std::strcmp(v, "ticket")) return 1;
    if (!std::strcmp(v, "tagged")) return 2;
    return *v ? 1 : 0;
}

void apply_environment(Tunables& t)
{
    for (const Entry& e : kEntries) {
        std::string var = "VB2_";
        for (const char* p = e.name; *p; ++p) var += (char)std::toupper((unsigned char)*p);
        if (const char* v = std::getenv(var.c_str())) t.*(e.field) = parse_value(v);
    }
    //  switches of earlier rounds that no code reads any more: said so, once -- set, they would A/B a configuration
    // against itself (tools/_run_c.sh did)
    static const char* const kRetired[] = {"VB2_SHARD_REDUCE", "VB2_DIGEST_CODES", "VB2_DICT_ORDER", "VB2_PAIRED", "VB2_GEOM",
                                           "VB2_GEOM_WAVES", "VB2_GEOM_GRID", "VB2_RELAY_REPS", "VB2_OWN_ROWS",
                                           "VB2_COHORT_SLOT_WG"};
    for (const char* name : kRetired)
        if (std::getenv(name))
            std::fprintf(stderr, "libvb2: environment variable %s is no longer read (csrc/tunables.h lists the switches)\n", name);
}

}  // namespace

Tunables& tunables()
{
    static Tunables t;
    static std::once_flag once;
    std::call_once(once, [] { apply_environment(t); });
    return t;
}

bool set_tunable(const char* name, int value)
{
    if (!name) return false;
    Tunables& t = tunables();
    for (const Entry& e : kEntries)
        if (!std::strcmp(e.name, name)) {
            t.*(e.field) = value;
            return true;
        }
    return false;
}

bool get_tunable(const char* name, int* value)
{
    if (!name || !value) return false;
    Tunables& t = tunables();
    for (const Entry& e : kEntries)
        if (!std::strcmp(e.name, name)) {
            *value = t.*(e.field);
            return true;
        }
    return false;
}

}  // namespace vb2
