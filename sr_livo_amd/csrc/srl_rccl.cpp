// srl_rccl.cpp -- run-time resolution of the process's single RCCL instance (see srl_rccl.h)
#include "srl_rccl.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

namespace {
SrlRccl g_tab;
bool g_ok = false;
std::string g_err;
std::once_flag g_once;
std::string g_override;          // srl_rccl_set_library: resolve from THIS shared object only (test stand-ins)
bool g_resolved = false;

template <class F>
bool sym(void *handle, const char *name, F &out) {
    void *p = dlsym(handle, name);
    out = reinterpret_cast<F>(p);
    return p != nullptr;
}

bool fill(void *handle) {
    sym(handle, "ncclCommCount", g_tab.CommCount);          // optional
    return sym(handle, "ncclGetVersion", g_tab.GetVersion) && sym(handle, "ncclGetUniqueId", g_tab.GetUniqueId) &&
           sym(handle, "ncclCommInitRank", g_tab.CommInitRank) && sym(handle, "ncclCommDestroy", g_tab.CommDestroy) &&
           sym(handle, "ncclAllReduce", g_tab.AllReduce) && sym(handle, "ncclAllGather", g_tab.AllGather) &&
           sym(handle, "ncclGetErrorString", g_tab.GetErrorString);
}

void resolve() {
    std::memset(&g_tab, 0, sizeof g_tab);
    g_resolved = true;
    if (!g_override.empty()) {
        void *h = dlopen(g_override.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h || !fill(h)) {
            const char *why = dlerror();
            g_err = "srl_comm_set_library(" + g_override + "): " + (why ? why : "an nccl entry point is missing");
            return;
        }
        g_tab.preloaded = false;
        int v = 0;
        if (g_tab.GetVersion(&v) != ncclSuccess) { g_err = "ncclGetVersion failed"; return; }
        g_tab.version = v;
        std::snprintf(g_tab.origin, sizeof g_tab.origin, "%s", g_override.c_str());
        g_ok = true;
        return;
    }
    // 1. an RCCL the process already carries (global symbol scope)
    if (dlsym(RTLD_DEFAULT, "ncclCommInitRank") && fill(RTLD_DEFAULT)) {
        g_tab.preloaded = true;
    } else {
        // python loads extension modules RTLD_LOCAL: torch's librccl may be mapped without being in the global scope.
        // RTLD_NOLOAD finds an already mapped object by SONAME without loading anything new.
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (h && fill(h)) {
            g_tab.preloaded = true;
        } else {
            const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
            for (const char *n : names) {
                h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (h && fill(h)) break;
                h = nullptr;
            }
            if (!h) {
                const char *why = dlerror();            // one call: dlerror() clears the message it returns
                g_err = std::string("no RCCL found: ") + (why ? why : "librccl.so.1 not on the library path");
                return;
            }
            g_tab.preloaded = false;
        }
    }
    int v = 0;
    if (g_tab.GetVersion(&v) != ncclSuccess) { g_err = "ncclGetVersion failed"; return; }
    g_tab.version = v;
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(g_tab.CommInitRank), &info) && info.dli_fname) std::snprintf(g_tab.origin, sizeof g_tab.origin, "%s", info.dli_fname);
    else std::snprintf(g_tab.origin, sizeof g_tab.origin, "unknown");
    g_ok = true;
}
}  // namespace

const SrlRccl *srl_rccl() {
    std::call_once(g_once, resolve);
    return g_ok ? &g_tab : nullptr;
}
const char *srl_rccl_error() { return g_err.c_str(); }
bool srl_rccl_set_library(const char *path) {
    if (g_resolved || !path || !*path) return false;       // before the first communicator call only: one RCCL instance per process
    g_override = path;
    return true;
}
