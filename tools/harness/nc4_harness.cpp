#include "mcf_nc4file.hpp"
#include <cstdio>
#include <random>
int main() {
    std::mt19937 rng(1);
    const int shapes[][3] = {{1, 1, 1}, {5, 7, 3}, {3, 300000, 2}, {5, 300000, 2}, {64, 33, 30}, {1, 2000000, 1}};
    for (auto& sh : shapes) {
        const int64_t R = sh[0], C = sh[1], T = sh[2];
        for (int level : {0, 1, 9}) {
            if (level == 9 && R * C > 100000) continue;
            mcf::Nc4File f;
            std::vector<double> east(C), north(R), tm(T);
            for (int64_t i = 0; i < C; ++i) east[i] = i;
            for (int64_t i = 0; i < R; ++i) north[i] = i;
            for (int64_t i = 0; i < T; ++i) tm[i] = i;
            std::vector<mcf::NcVarDef> vars = {{"Tz", "Air temperature", "deg C x 100"}, {"soilm", "Soil surface moisture", "x"}};
            std::string e = f.create4("/tmp/asan_t.nc", R, C, T, east.data(), north.data(), tm.data(), "wkt", vars, level);
            if (!e.empty()) { printf("create: %s\n", e.c_str()); return 1; }
            std::vector<uint8_t> recs((size_t)(T * f.rec_bytes));
            for (auto& b : recs) b = (uint8_t)(rng() & 3);
            // out of order, in two pieces
            const int64_t h = T / 2;
            e = f.write_records(h, T - h, recs.data() + h * f.rec_bytes);
            if (e.empty() && h > 0) e = f.write_records(0, h, recs.data());
            if (e.empty()) e = f.write_records(0, 0, recs.data());
            if (!e.empty()) { printf("write: %s\n", e.c_str()); return 1; }
            std::string bad = f.write_records(T, 1, recs.data());
            if (bad.empty()) { printf("range not checked\n"); return 1; }
            e = f.close();
            if (!e.empty()) { printf("close: %s\n", e.c_str()); return 1; }
            printf("%lldx%lldx%lld level %d ok\n", (long long)R, (long long)C, (long long)T, level);
        }
    }
    mcf::Nc4File g;
    std::vector<mcf::NcVarDef> v1 = {{"Tz", "a", "b"}};
    double one = 1;
    std::string e = g.create4("/tmp/no_such_dir_x/y.nc", 1, 1, 1, &one, &one, &one, nullptr, v1, 9);
    printf("bad path: %s\n", e.c_str());
    return e.empty();
}
