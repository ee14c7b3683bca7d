// cpu_check.cpp -- the host-side logic that needs no GPU, as one program: az::parse_gpus (AZ_GPUS of the
// generator), the per-GPU worker thread of the group API, and -- when linked against libazhip.so -- the error
// paths of az_group_create that return before any GPU is touched.  Built twice more by `make sanitize` with
// -fsanitize=address,undefined and -fsanitize=thread (those builds leave the library out: -DAZ_CPU_CHECK_NO_LIB).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../csrc/az_worker.h"
#include "../include/az/selfplay.hpp"

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

static bool throws(const char* spec, int n_visible = -1) {
    try { az::parse_gpus(spec, 0, n_visible); } catch (const az::Error&) { return true; }
    return false;
}

static thread_local std::string t_err;
static const char* fake_last_error() { return t_err.c_str(); }

int main() {
    using V = std::vector<int>;
    EXPECT(az::parse_gpus(nullptr, 3) == V{3});
    EXPECT(az::parse_gpus("", 2) == V{2});
    EXPECT(az::parse_gpus("1", 5) == V{0});                       // a count: GPUs 0 .. n-1
    EXPECT((az::parse_gpus("4", 0) == V{0, 1, 2, 3}));
    EXPECT((az::parse_gpus("all", 0, 8) == V{0, 1, 2, 3, 4, 5, 6, 7}));
    EXPECT((az::parse_gpus("2,5,7", 0) == V{2, 5, 7}));
    EXPECT((az::parse_gpus("3,1", 0) == V{3, 1}));                 // order kept: the first GPU receives the blob
    EXPECT(throws("all"));                                        // needs the number of visible GPUs
    EXPECT(throws("0"));
    EXPECT(throws("-2"));
    EXPECT(throws("x"));
    EXPECT(throws("1,,2"));
    EXPECT(throws("1,2,"));
    // AZ_NET_CHANNELS: NetConfig 2 when unset, <channels>x<entries>, or a list
    EXPECT(az::parse_channels(nullptr) == az::NetConfig::get(2).resblocks());
    EXPECT(az::parse_channels("") == az::NetConfig::get(2).resblocks());
    EXPECT((az::parse_channels("256x21") == V(21, 256)));
    EXPECT((az::parse_channels("32,64,64") == V{32, 64, 64}));
    for (const char* bad : {"256x1", "0x4", "x", "128x65", "64", "64,,64", "64,-1"}) {
        bool threw = false;
        try { az::parse_channels(bad); } catch (const az::Error&) { threw = true; }
        EXPECT(threw);
    }

    // AZ_PRECISION
    EXPECT(az::parse_precision(nullptr) == AZ_PREC_FP16 && az::parse_precision("") == AZ_PREC_FP16 && az::parse_precision("fp16") == AZ_PREC_FP16);
    EXPECT(az::parse_precision("bf16") == AZ_PREC_BF16 && az::parse_precision("fp16x3") == AZ_PREC_FP16X3);
    for (const char* bad : {"fp32", "FP16", "x3", "fp16x"}) {
        bool threw = false;
        try { az::parse_precision(bad); } catch (const az::Error&) { threw = true; }
        EXPECT(threw);
    }

    // workers: jobs run in order on their own thread, statuses and per-thread error texts come back
    {
        std::vector<az_worker::Worker*> ws;
        for (int i = 0; i < 4; ++i) ws.push_back(new az_worker::Worker(fake_last_error));
        std::atomic<long> sum{0};
        unsigned long seq[4] = {0, 0, 0, 0};
        for (int round = 0; round < 200; ++round) {
            for (int i = 0; i < 4; ++i)
                ws[i]->post([&, i, round] {
                    seq[i] = seq[i] * 3 + (unsigned long)round;                 // only worker i touches seq[i]
                    sum += i + 1;
                    if (round % 50 == 49 && i == 2) { t_err = "job " + std::to_string(round) + " failed"; return -4; }
                    return 0;
                });
            for (int i = 0; i < 4; ++i) {
                std::string m;
                const int rc = ws[i]->wait(m);
                if (round % 50 == 49 && i == 2) { EXPECT(rc == -4); EXPECT(m == "job " + std::to_string(round) + " failed"); }
                else { EXPECT(rc == 0); EXPECT(m.empty()); }
            }
        }
        EXPECT(sum == 200L * 10);
        unsigned long want = 0;
        for (int round = 0; round < 200; ++round) want = want * 3 + (unsigned long)round;
        for (int i = 0; i < 4; ++i) EXPECT(seq[i] == want);
        for (auto* w : ws) delete w;                              // joins idle threads
        delete new az_worker::Worker(fake_last_error);            // a worker that never got a job
    }

#ifndef AZ_CPU_CHECK_NO_LIB
    // az_group_create: argument errors come back before any GPU is touched, nothing is left behind
    {
        az_config cfg{};
        cfg.N = 9; cfg.K = 5; cfg.G = 4; cfg.eval_mode = AZ_EVAL_HASH; cfg.max_budget = 8;
        az_group* g = (az_group*)0x1;
        const int dup[3] = {0, 1, 0};
        EXPECT(az_group_create(&g, &cfg, dup, 3, 1) == AZ_E_ARG && g == nullptr);
        EXPECT(std::strstr(az_last_error(), "listed twice") != nullptr);
        const int one[1] = {0};
        EXPECT(az_group_create(&g, &cfg, one, 0, 1) == AZ_E_ARG);
        EXPECT(az_group_create(&g, &cfg, one, 65, 1) == AZ_E_ARG);
        EXPECT(az_group_create(&g, &cfg, one, 1, 0) == AZ_E_ARG);
        EXPECT(az_group_create(&g, &cfg, one, 1, 9) == AZ_E_ARG);
        EXPECT(az_group_create(nullptr, &cfg, one, 1, 1) == AZ_E_ARG);
        EXPECT(az_group_create(&g, nullptr, one, 1, 1) == AZ_E_ARG);
        // RCCL that cannot be loaded: a group of two GPUs fails cleanly with AZ_E_HIP and says why ...
        setenv("AZ_RCCL_LIB", "/nonexistent/librccl.so.1", 1);
        const int two[2] = {0, 1};
        EXPECT(az_group_create(&g, &cfg, two, 2, 1) == AZ_E_HIP && g == nullptr);
        EXPECT(std::strstr(az_last_error(), "librccl not found") != nullptr);
        // ... and a group of ONE GPU does not ask for RCCL at all: what stops it here is the missing GPU (on a GPU
        // box it is created; tests/test_gpu_shim.py runs the generator that way)
        const int rc = az_group_create(&g, &cfg, one, 1, 1);
        if (az_device_count() < 1) {
            EXPECT(rc == AZ_E_HIP && g == nullptr);
            EXPECT(std::strstr(az_last_error(), "librccl") == nullptr);
        } else {
            EXPECT(rc == AZ_OK && g != nullptr && az_group_contexts(g) == 1);
            az_group_destroy(g);
            const int far[1] = {az_device_count() + 3};           // a device the process does not have
            EXPECT(az_group_create(&g, &cfg, far, 1, 1) == AZ_E_ARG && g == nullptr);
        }
        EXPECT(az_group_destroy(nullptr) == AZ_OK);
    }
#endif
    std::printf(fails ? "cpu_check: %d FAILED\n" : "cpu_check ok\n", fails);
    return fails ? 1 : 0;
}
