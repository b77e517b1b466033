// CPU unit test of host/device_replicas.h (the per-device replica table of mvfx_cube_lut) with fake device ordinals -- no GPU, no HIP.
//   g++ -std=c++17 -O1 -pthread -Igst-plugin-rs_amd/host tests/replica_table_test.cpp -o /tmp/replica_table_test && /tmp/replica_table_test
#include "device_replicas.h"

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

struct Copy {
    int device;
    static std::atomic<int> alive, made;
    explicit Copy(int d) : device(d) { alive++; made++; }
    ~Copy() { alive--; }
};
std::atomic<int> Copy::alive{0}, Copy::made{0};

#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main()
{
    {
        mvfx::DeviceReplicas<Copy, 8> t;
        CHECK(t.count() == 0 && t.find(0) == nullptr && t.find(-1) == nullptr && t.find(8) == nullptr);
        bool created = false;
        Copy *a = t.get_or_create(3, [] { return new Copy(3); }, &created);
        CHECK(a && created && a->device == 3 && t.count() == 1 && t.find(3) == a);
        // a second user on the same device shares the replica: no second make()
        Copy *b = t.get_or_create(3, [] { return new Copy(-99); }, &created);
        CHECK(b == a && !created && Copy::made == 1);
        // another device gets its own; the first one stays (round 5 freed and re-uploaded on a switch)
        Copy *c = t.get_or_create(0, [] { return new Copy(0); }, &created);
        CHECK(c && c != a && created && t.count() == 2 && t.find(3) == a && t.find(0) == c);
        // switching back and forth makes nothing new
        for (int i = 0; i < 100; i++) CHECK(t.get_or_create(i & 1 ? 3 : 0, [] { return new Copy(-1); }) == (i & 1 ? a : c));
        CHECK(Copy::made == 2);
        // out of range ordinals and a failing make() yield nullptr and leave no slot behind
        CHECK(t.get_or_create(8, [] { return new Copy(8); }) == nullptr && t.get_or_create(-1, [] { return new Copy(-1); }) == nullptr && Copy::made == 2);
        CHECK(t.get_or_create(5, []() -> Copy * { return nullptr; }, &created) == nullptr && !created && t.find(5) == nullptr && t.count() == 2);
        int seen = 0, order = -1;
        bool ordered = true;
        t.for_each([&](int d, Copy &r) { seen++; ordered = ordered && d > order && r.device == d; order = d; });
        CHECK(seen == 2 && ordered);
        // sixteen threads asking for the same new device: ONE make(), everybody gets its result
        std::vector<std::thread> th;
        std::vector<Copy *> got(16, nullptr);
        for (int i = 0; i < 16; i++) th.emplace_back([&, i] { got[i] = t.get_or_create(7, [] { return new Copy(7); }); });
        for (auto &x : th) x.join();
        for (int i = 0; i < 16; i++) CHECK(got[i] && got[i] == got[0]);
        CHECK(Copy::made == 3 && Copy::alive == 3 && t.count() == 3);
    }
    CHECK(Copy::alive == 0); // the table owns its replicas
    std::printf("replica table ok\n");
    return 0;
}
