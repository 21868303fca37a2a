// pgps_dyn.cpp -- see pgps_dyn.h.  Host code only.
#include "pgps_dyn.h"

#include <dlfcn.h>

#include <mutex>

namespace pgps {
namespace dyn {
namespace {

// the copy the process already has, else the one the loader finds, else ROCm's default place
void* open_lib(const char* const* names, std::string* which) {
    for (int pass = 0; pass < 2; ++pass) {
        for (const char* const* n = names; *n; ++n) {
            void* h = dlopen(*n, pass == 0 ? (RTLD_NOW | RTLD_NOLOAD) : (RTLD_NOW | RTLD_LOCAL));
            if (h) {
                if (which) *which = std::string(*n) + (pass == 0 ? " (already in the process)" : "");
                return h;
            }
        }
    }
    return nullptr;
}

template <typename F>
bool sym(void* h, const char* name, F* out, std::string* err) {
    *out = reinterpret_cast<F>(dlsym(h, name));
    if (!*out && err->empty()) *err = std::string("missing symbol ") + name;
    return *out != nullptr;
}

}  // namespace

const Rccl& rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        static const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
        void* h = open_lib(names, &api.path);
        if (!h) {
            const char* e = dlerror();
            api.err = std::string("librccl.so.1 not found") + (e ? std::string(": ") + e : std::string());
            return;
        }
        bool all = true;
        all &= sym(h, "ncclGetUniqueId", &api.GetUniqueId, &api.err);
        all &= sym(h, "ncclCommInitRank", &api.CommInitRank, &api.err);
        all &= sym(h, "ncclCommDestroy", &api.CommDestroy, &api.err);
        all &= sym(h, "ncclCommCount", &api.CommCount, &api.err);
        all &= sym(h, "ncclCommUserRank", &api.CommUserRank, &api.err);
        all &= sym(h, "ncclAllGather", &api.AllGather, &api.err);
        all &= sym(h, "ncclGetErrorString", &api.GetErrorString, &api.err);
        api.ok = all;
    });
    return api;
}

namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
const Roctx& roctx() {
    static Roctx api;
    static std::once_flag once;
    std::call_once(once, [] {
        static const char* const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                                            "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", nullptr};
        void* h = open_lib(names, nullptr);
        if (!h) return;
        std::string err;
        if (!sym(h, "roctxRangePushA", &api.push, &err) || !sym(h, "roctxRangePop", &api.pop, &err)) api = Roctx{};
    });
    return api;
}
}  // namespace

void range_push(const char* name) {
    const Roctx& r = roctx();
    if (r.push) r.push(name);
}
void range_pop() {
    const Roctx& r = roctx();
    if (r.pop) r.pop();
}

}  // namespace dyn
}  // namespace pgps
