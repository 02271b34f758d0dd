#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>
#include "femo_symbolic.h"
template <class T> std::vector<T> rd(const std::string& f, size_t n) { std::vector<T> v(n); FILE* h = fopen(f.c_str(), "rb"); if (!h || fread(v.data(), sizeof(T), n, h) != n) { printf("read %s failed\n", f.c_str()); exit(2); } fclose(h); return v; }
int main() {
    for (const char* name : {"quad", "tri"}) {
        std::string b = std::string("/tmp/sym_") + name;
        int nel, nP2, nV, npc, ndpc; FILE* h = fopen((b + "_meta.txt").c_str(), "r"); if (fscanf(h, "%d %d %d %d %d", &nel, &nP2, &nV, &npc, &ndpc) != 5) return 3; fclose(h);
        auto p2 = rd<int32_t>(b + "_p2.bin", (size_t)nel * npc); auto cent = rd<double>(b + "_cent.bin", (size_t)nel * 3); auto cext = rd<double>(b + "_cext.bin", (size_t)nel * 3);
        auto dofs = rd<int32_t>(b + "_dofs.bin", (size_t)nel * ndpc);
        for (int rule = 0; rule <= 2; ++rule)
            for (int leaf : {1, 5, 12, 24})
                for (int md : {0, 3}) {
                    femo_plan* p = nullptr;
                    int rc = femo_plan_build_ex(&p, nel, nP2, nV, npc, ndpc, p2.data(), cent.data(), cext.data(), dofs.data(), leaf, md, rule, rule ? 0.75 : 0.0);
                    if (rc) { printf("%s rule %d leaf %d depth %d: rc %d %s\n", name, rule, leaf, md, rc, femo_plan_last_error()); return 1; }
                    long long n = femo_plan_size(p, "nf");
                    std::vector<int32_t> nf(n); femo_plan_get(p, "nf", nf.data(), n * 4);
                    femo_plan_free(p);
                }
        printf("%s: ok\n", name);
    }
    return 0;
}
