// CPU harness for active-gym_amd/csrc/agx_device_guard.h with a mocked device runtime (no HIP): the ordinal != 0 cases an
// 8-GPU node has and a one-GPU test box cannot run.  Prints "ok" or the first failed check.
#include <cstdio>
#include <vector>

#include "agx_device_guard.h"

static int g_cur = 0, g_ndev = 8;
static bool g_get_fails = false, g_set_fails = false;
static std::vector<int> g_sets;
struct MockApi {
    static int get(int *d) {
        if (g_get_fails) return 1;
        *d = g_cur;
        return 0;
    }
    static int set(int d) {
        g_sets.push_back(d);
        if (g_set_fails || d < 0 || d >= g_ndev) return 1;
        g_cur = d;
        return 0;
    }
};
using Guard = agx::DeviceGuardT<MockApi>;
#define CHECK(c)                                  \
    do {                                          \
        if (!(c)) {                               \
            printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
            return 1;                             \
        }                                         \
    } while (0)

int main() {
    // a context on device 3 used from a thread on device 0: switched inside the scope, restored after
    g_cur = 0;
    {
        Guard g(3);
        CHECK(g.ok && g.switched && g.prev == 0 && g_cur == 3);
    }
    CHECK(g_cur == 0 && g_sets.size() == 2 && g_sets[0] == 3 && g_sets[1] == 0);
    // same device: no runtime call at all
    g_sets.clear();
    g_cur = 5;
    {
        Guard g(5);
        CHECK(g.ok && !g.switched && g_cur == 5);
    }
    CHECK(g_cur == 5 && g_sets.empty());
    // nested entry points (agx_step_fixed -> agx_ingest -> ...): inner guards see the switched device and do nothing
    g_sets.clear();
    g_cur = 1;
    {
        Guard a(6);
        {
            Guard b(6);
            CHECK(!b.switched && g_cur == 6);
            {
                Guard c(2);                                      // another context on another device, mid-call
                CHECK(c.switched && g_cur == 2);
            }
            CHECK(g_cur == 6);
        }
        CHECK(g_cur == 6);
    }
    CHECK(g_cur == 1);
    // the context's device cannot be selected: ok is false, the caller's device untouched, nothing "restored"
    g_sets.clear();
    g_cur = 0;
    {
        Guard g(99);
        CHECK(!g.ok && !g.switched && g_cur == 0);
    }
    CHECK(g_cur == 0 && g_sets.size() == 1);
    // the caller's device cannot be read: the switch still happens, nothing is restored (there is nothing known to restore)
    g_sets.clear();
    g_cur = 4;
    g_get_fails = true;
    {
        Guard g(7);
        CHECK(g.ok && g.switched && g.prev == -1 && g_cur == 7);
    }
    CHECK(g_cur == 7 && g_sets.size() == 1);
    g_get_fails = false;
    // a failing restore does not throw or loop
    g_sets.clear();
    g_cur = 2;
    {
        Guard g(3);
        g_set_fails = true;
    }
    g_set_fails = false;
    CHECK(g_cur == 3 && g_sets.size() == 2);
    printf("ok\n");
    return 0;
}
