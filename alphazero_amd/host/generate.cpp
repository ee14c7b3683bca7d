// generate.cpp -- this build's self-play producer, the drop-in for the
// reference's `./generate <nthreads> <iter_budget> <verbosity>`
// (generate.cpp:37-123): same argv, same ckpt_location.txt weight hand-off,
// same PUSH stream on tcp://*:5555 in the trainer's 3-frame format, "press
// enter" to stop (generate.cpp:19-23).  Instead of one game searched by
// nthreads CPU threads it advances AZ_GAMES (default 512) games per GPU side by side
// on every GPU named by AZ_GPUS ("all", a count, or a list; default: AZ_DEVICE or 0),
// one host thread per GPU; new weights reach the GPUs by one RCCL broadcast; all rows
// leave through the one PUSH socket.  <nthreads> is accepted and ignored.
#include <atomic>
#include <chrono>
#include <iostream>
#include <thread>

#include "../include/az/env.hpp"
#include "../include/az/replay.hpp"
#include "../include/az/selfplay.hpp"

static std::atomic<bool> alive{true};
enum { GEN, TRN };
using Env = az::Gomoku;

static void run() {
    std::cin.get();
    alive = false;
}

int main(int argc, char const* argv[]) {
    if (argc != 4)
        throw std::runtime_error("Invalid arguments. Usage: ./generate <nthreads> <iter_budget> <verbosity>");
    std::thread thd(&run);
    const int iter_budget = atoi(argv[2]);
    const int verbosity = atoi(argv[3]);
    const char* g = getenv("AZ_GAMES");
    const char* dev = getenv("AZ_DEVICE");
    const char* port = getenv("AZ_PORT");
    const char* max_s = getenv("AZ_SECONDS");
    const int games = g ? atoi(g) : 512;

    const std::vector<int> devices = az::parse_gpus(getenv("AZ_GPUS"), dev ? atoi(dev) : 0, az_device_count());
    az::SelfPlay<Env> agent(games, iter_budget, 3, 3, devices[0], std::random_device{}(), 0, 0, devices,
                            az::parse_channels(getenv("AZ_NET_CHANNELS")), az::parse_precision(getenv("AZ_PRECISION")));
    std::cout << "GPUs: " << devices.size() << " | games per GPU: " << games << std::endl;
    az::ReplayBuffer<Env> buffer(GEN, "*", port ? port : "5555");
    agent.load();
    // Every move's rounds as ONE launch of the persistent tower / tree pair, with simulations that end in a stored
    // terminal leaf (no network call, mcts.hpp:232-233) riding along in a round: the producer's default wherever the
    // contexts can run it (the one-launch tower, a GPU to themselves).  A short trial says whether they can; if
    // not, or with AZ_PAIR=0, the rounds are lockstep launches.
    bool pair = false;
    if (!getenv("AZ_PAIR")) {
        setenv("AZ_PAIR", "1", 1);
        setenv("AZ_CHAIN", "6", 0);
        try {
            agent.begin();
            agent.run_pair(8);
            (void)agent.stats();
            pair = true;
        } catch (const az::Error& e) {
            setenv("AZ_PAIR", "0", 1);
            std::cout << "The persistent tower / tree pair does not run here (" << e.what() << "); using lockstep rounds." << std::endl;
        }
    } else {
        pair = atoi(getenv("AZ_PAIR")) != 0;
    }
    agent.begin();
    const auto t0 = std::chrono::steady_clock::now();
    long long rows = 0;
    auto emit = [&](const az::ByteTensor& board, const az::FloatTensor& policy, const az::FloatTensor& reward) {
        buffer.temporary_append(board, policy);
        buffer.send_reward(reward);
    };
    // The GPUs never wait for the host: while the rows of move k are expanded into their 8 symmetries and sent,
    // move k + 1 of every game is already queued.  AZ_NO_OVERLAP=1 keeps the plain order (run, wait, drain) for A/B.
    const bool overlap = getenv("AZ_NO_OVERLAP") == nullptr;
    agent.run(iter_budget);                       // about one move per slot
    while (alive) {
        try {
            if (overlap) {
                agent.flip();                     // rows of the rounds queued so far: the retired rings
                agent.run(iter_budget);           // the next move, queued behind them
                rows += agent.drain_retired(emit);    // waits for the retired rounds only
            } else {
                rows += agent.drain(emit);
                agent.run(iter_budget);
            }
        } catch (const az::Error& e) {
            // A pair that stops being resident together in the middle of a run (another process arriving on the GPU)
            // leaves its games out of step: start them again as lockstep rounds and go on producing.
            if (!pair || std::string(e.what()).find("hand-over wait") == std::string::npos) throw;
            std::cout << "The persistent tower / tree pair stopped running (" << e.what() << "); games restarted as lockstep rounds." << std::endl;
            setenv("AZ_PAIR", "0", 1);
            pair = false;
            agent.begin();
            agent.run(iter_budget);
        }
        agent.load();                             // agent.load() per game start in the reference (generate.cpp:73)
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (verbosity >= 1) {
            const az_stats s = agent.stats();     // (waits for the queued rounds: verbosity 0 is the measured mode)
            std::cout << "Game count: " << s.games << " | moves: " << s.moves << " | sims/s: " << (double)s.sims / el
                      << " | rows sent: " << rows * 8 << std::endl;
        }
        if (max_s && el > atof(max_s)) break;
    }
    {   // whole-run rates, one JSON line (AZ_SECONDS runs; the interactive run ends with "press enter")
        rows += agent.drain(emit);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const az_stats s = agent.stats();
        std::cout << "{\"generate\": {\"seconds\": " << el << ", \"sims_per_s\": " << (double)s.sims_backed_up / el
                  << ", \"moves_per_s\": " << (double)s.moves / el << ", \"games\": " << s.games
                  << ", \"rows_per_s\": " << (double)rows / el << ", \"messages_per_s\": " << (double)rows * 8 / el
                  << ", \"overlap\": " << (overlap ? "true" : "false") << ", \"persistent_pair\": " << (pair ? "true" : "false")
                  << ", \"gpus\": " << devices.size()
                  << ", \"games_per_gpu\": " << games << ", \"replay_dropped\": " << s.replay_dropped << "}}" << std::endl;
    }
    alive = false;
    thd.detach();
    return 0;
}
