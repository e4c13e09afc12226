// nbody_run.cpp -- thin C++ host over the C ABI: the reference's main loop (main_project/kernel.cu:1067-1295)
// without the window.  Load or generate bodies, initialize(), setParticlesPosition/Velocity(), then one
// step() per "frame"; instead of drawing, it reports timing, energy and momentum and can dump / resume raw
// snapshots (the reference keeps its state only in the VBO).
//
//   nbody_run [dataset_id]                               the reference's CLI (kernel.cu:1069-1086), files in --data-dir
//   nbody_run --plummer 65536 --steps 100 --dt 1e-3 --softening 1e-3 --energy-every 10
//   nbody_run --file galaxy.bin --steps 1000 --dump-every 100 --dump-prefix out/gal
//   nbody_run --resume out/gal_000500.nbs --steps 500
//   nbody_run --plummer 1048576 --devices 0,1,2,3,4,5,6,7 --pair-once --steps 10     rows sharded over 8 GPUs, RCCL inside
//   mpirun -np 8 nbody_run --plummer 4194304 --ranks-from-env --id-file /tmp/nbody.id --auto --steps 1000 --energy-every 100
//                                                        one process per GPU (rank / world size / local rank from the launcher's
//                                                        environment, or --rank R --world P [--device D]); rank 0 reports
//
// Build: g++ -O2 -std=c++17 -Iinclude host/nbody_run.cpp -Ln_body_problem_amd -lnbody_amd -Wl,-rpath,'$ORIGIN/../n_body_problem_amd'
#include "../include/nbody.hpp"
#include "nbody_io.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

static void usage()
{
    std::cerr << "usage: nbody_run [dataset_id 0-5] [--data-dir DIR] [--file PATH] [--plummer N] [--seed S] [--resume SNAPSHOT]\n"
                 "                 [--steps K] [--dt DT] [--softening EPS] [--energy-every M] [--dump-every M] [--dump-prefix P] [--morton [--reorder-every M]]\n"
                 "                 [--pad-reference] [--device D] [--final SNAPSHOT] [--kdk] [--pair-once | --auto] [--particle-softening]\n"
                 "                 [--devices D0,D1,...] [--ring] [--peer-copy]      rows sharded over several GPUs (library-owned exchange)\n"
                 "                 [--rank R --world P | --ranks-from-env] --id-file PATH   the same with one process per GPU: rank 0 writes the\n"
                 "                                  communicator id to PATH (a path every rank sees), the others wait for it; --device D or LOCAL_RANK\n";
}

// The first of these variables that is set, as an integer (launchers: torchrun, Open MPI, MPICH / Slurm).
static int env_int(std::initializer_list<const char *> names, int fallback)
{
    for (const char *n : names)
        if (const char *v = std::getenv(n))
            return std::atoi(v);
    return fallback;
}

// Rank 0 creates the communicator id and publishes it through a file (written beside it, then renamed: never seen half
// written); the other ranks wait for the file.
static std::vector<unsigned char> exchange_id(int rank, const std::string &path)
{
    if (path.empty()) throw std::runtime_error("one process per GPU needs --id-file PATH (a path all ranks see)");
    if (rank == 0) {
        if (std::ifstream(path).good())  // a leftover of another run would be read by the ranks that start before this one
            throw std::runtime_error(path + " exists already: remove it or give this run its own --id-file");
        const std::vector<unsigned char> id = nbody::MultiSystem::uniqueId();
        const std::string tmp = path + ".tmp";
        { std::ofstream f(tmp, std::ios::binary | std::ios::trunc); f.write(reinterpret_cast<const char *>(id.data()), (std::streamsize)id.size()); }
        if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot write " + path);
        return id;
    }
    for (int waited_ms = 0; waited_ms < 120000; waited_ms += 10) {
        std::ifstream f(path, std::ios::binary);
        std::vector<unsigned char> id((size_t)NBODY_UNIQUE_ID_BYTES);
        if (f && f.read(reinterpret_cast<char *>(id.data()), (std::streamsize)id.size())) return id;
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    throw std::runtime_error("no communicator id in " + path + " after 120 s (is rank 0 running?)");
}

int main(int argc, char **argv)
{
    int dataset = -1, device = 0;
    std::string data_dir = "./data", file, resume, dump_prefix = "nbody", final_path;
    std::int64_t plummer_n = 0, steps = 100, energy_every = 0, dump_every = 0, reorder_every = 0;
    std::uint64_t seed = 0x5EED0003ull;
    float dt = nbody::kTimeTick, softening = nbody::kSofteningVersion3;  // the reference's constants
    bool pad = false, kdk = false, pair_once = false, particle_eps = false, ring = false, peer_copy = false, morton = false;
    bool auto_mode = false;  // --auto: the library picks the force mode by body count (nbody_create_auto / NBODY_FORCE_AUTO)
    std::vector<int> devices;
    int rank = 0, world = 1;   // one process per GPU: --rank / --world or the launcher's environment
    bool device_given = false;
    std::string id_file;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "--data-dir") data_dir = next();
        else if (a == "--file") file = next();
        else if (a == "--plummer") plummer_n = std::atoll(next().c_str());
        else if (a == "--seed") seed = std::strtoull(next().c_str(), nullptr, 0);
        else if (a == "--resume") resume = next();
        else if (a == "--steps") steps = std::atoll(next().c_str());
        else if (a == "--dt") dt = (float)std::atof(next().c_str());
        else if (a == "--softening") softening = (float)std::atof(next().c_str());
        else if (a == "--energy-every") energy_every = std::atoll(next().c_str());
        else if (a == "--dump-every") dump_every = std::atoll(next().c_str());
        else if (a == "--dump-prefix") dump_prefix = next();
        else if (a == "--final") final_path = next();
        else if (a == "--pad-reference") pad = true;
        else if (a == "--kdk") kdk = true;
        else if (a == "--pair-once") pair_once = true;
        else if (a == "--auto") auto_mode = true;
        else if (a == "--particle-softening") particle_eps = true;
        else if (a == "--device") { device = std::atoi(next().c_str()); device_given = true; }
        else if (a == "--rank") rank = std::atoi(next().c_str());
        else if (a == "--world") world = std::atoi(next().c_str());
        else if (a == "--id-file") id_file = next();
        else if (a == "--ranks-from-env") {
            rank = env_int({"RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID"}, 0);
            world = env_int({"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS"}, 1);
            if (!device_given) device = env_int({"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"}, 0);
        }
        else if (a == "--devices") {
            std::stringstream list(next());
            for (std::string tok; std::getline(list, tok, ',');) devices.push_back(std::atoi(tok.c_str()));
        }
        else if (a == "--ring") ring = true;
        else if (a == "--peer-copy") peer_copy = true;
        else if (a == "--morton") morton = true;
        else if (a == "--reorder-every") reorder_every = std::atoll(next().c_str());  // with --morton: refresh the layout
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (!a.empty() && a[0] != '-') {
            dataset = std::atoi(a.c_str());  // kernel.cu:1069-1086: argv[1] = dataset id, 0..5
            if (dataset < 0 || dataset > 5) { std::cerr << "dataset id must be 0..5\n"; return 2; }
        } else { usage(); return 2; }
    }
    if (world < 1 || rank < 0 || rank >= world) { std::cerr << "--rank must lie in [0, --world)\n"; return 2; }
    if (world > 1 && !devices.empty()) { std::cerr << "--devices (one process, several GPUs) and --world (one process per GPU) exclude each other\n"; return 2; }
    const bool per_process = world > 1;
    const bool report = rank == 0;          // one process per GPU: every rank computes, rank 0 prints and writes files
    try {
        nbody_io::Bodies b;
        std::int64_t step0 = 0;
        double time0 = 0.0;
        if (!resume.empty()) b = nbody_io::load_snapshot(resume, &step0, &time0);
        else if (!file.empty()) b = nbody_io::read_any(file);
        else if (dataset >= 0) b = nbody_io::read_any(data_dir + "/" + nbody_io::reference_dataset(dataset));
        else b = nbody_io::plummer(plummer_n > 0 ? plummer_n : 65536, seed);
        const std::int64_t n_real = b.n();
        if (pad) nbody_io::pad_reference_style(b);  // accepted, never required (kernel.cu:260-278)
        if (report)
            std::printf("num of Bodies = %lld (real %lld)  dt = %g  softening = %g\n", (long long)b.n(), (long long)n_real, dt, softening);
        // --morton: the bodies are stored along a Morton curve (nbody_morton_order: neighbours in memory are neighbours in
        // space, the force kernels' operands toggle fewer bits, the power-limited clock rises); snapshots keep the file's order
        std::vector<std::int64_t> order;
        if (morton && b.n() > 0 && devices.empty() && !per_process) {  // (sharded: the library does it, nbody_multi_config.body_order)
            order.resize((size_t)b.n());
            if (nbody_morton_order(b.pos.data(), b.n(), order.data()) != NBODY_OK) throw std::runtime_error("nbody_morton_order failed");
            nbody_io::Bodies sorted = b;
            for (std::int64_t k = 0; k < b.n(); ++k)
                for (int c = 0; c < 4; ++c) {
                    sorted.pos[4 * (size_t)k + c] = b.pos[4 * (size_t)order[(size_t)k] + c];
                    sorted.vel[4 * (size_t)k + c] = b.vel[4 * (size_t)order[(size_t)k] + c];
                }
            b = sorted;
        }
        auto save = [&](const std::string &name, std::int64_t step, double time) {  // b holds the device's order
            if (order.empty()) { nbody_io::save_snapshot(name, b, step, time); return; }
            nbody_io::Bodies out = b;
            for (std::int64_t k = 0; k < b.n(); ++k)
                for (int c = 0; c < 4; ++c) {
                    out.pos[4 * (size_t)order[(size_t)k] + c] = b.pos[4 * (size_t)k + c];
                    out.vel[4 * (size_t)order[(size_t)k] + c] = b.vel[4 * (size_t)k + c];
                }
            nbody_io::save_snapshot(name, out, step, time);
        };

        if (!devices.empty() || per_process) {  // rows sharded over GPUs; everything per step happens inside the library
            nbody::MultiSystem ms;
            if (per_process)   // this process is one rank of `world`; every call below is collective
                ms.initializeRank(b.n(), device, rank, world, exchange_id(rank, id_file), pair_once, kdk, ring, 0, morton, auto_mode);
            else
                ms.initialize(b.n(), devices, pair_once, kdk, ring, peer_copy, 0, morton, auto_mode);
            ms.setState(b.pos.data(), b.vel.data());
            if (morton && reorder_every > 0) ms.setReorderPeriod(reorder_every);
            ms.timing(true);
            if (particle_eps) {
                std::vector<float> eps((size_t)b.n());
                for (std::int64_t i = 0; i < b.n(); ++i) eps[(size_t)i] = b.vel[4 * (size_t)i + 3];
                ms.setParticleSoftening(eps.data());
            }
            auto inf = ms.info();
            if (report)
                std::printf("ranks = %lld (RCCL communicator of %lld)  padded bodies = %lld  rows per rank = %lld  split = %lld\n",
                        (long long)inf[4], (long long)inf[6], (long long)inf[1], (long long)inf[2], (long long)inf[3]);
            nbody::System::Energy e0{};
            if (energy_every > 0) {
                e0 = ms.energy(softening);
                if (report)
                    std::printf("step %lld  E = %.9e (K %.6e U %.6e)\n", (long long)step0, e0.total, e0.kinetic, e0.potential);
            }
            const auto t0 = std::chrono::steady_clock::now();
            std::int64_t s = 0;
            while (s < steps) {  // steps between two reports are enqueued back to back: the exchange stays in flight
                std::int64_t k = steps - s;
                if (energy_every > 0) k = std::min(k, energy_every - s % energy_every);
                if (dump_every > 0) k = std::min(k, dump_every - s % dump_every);
                ms.stepN((int)k, dt, softening);
                s += k;
                const std::int64_t gs = step0 + s;
                if (energy_every > 0 && (s % energy_every == 0 || s == steps)) {
                    auto e = ms.energy(softening);
                    auto p = ms.momentum();
                    if (report)
                        std::printf("step %lld  E = %.9e  dE/E0 = %+.3e  |p| = %.3e\n", (long long)gs, e.total,
                                (e.total - e0.total) / std::fabs(e0.total), std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
                }
                if (dump_every > 0 && s % dump_every == 0) {
                    ms.download(b.pos.data(), b.vel.data());  // collective; every rank receives all rows
                    char name[512];
                    std::snprintf(name, sizeof name, "%s_%06lld.nbs", dump_prefix.c_str(), (long long)gs);
                    if (report)
                        save(name, gs, time0 + (double)s * dt);
                }
            }
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double inter = (double)b.n() * (double)b.n() * (double)steps;
            const bool identical = ms.replicasIdentical();  // collective
            if (report)
                std::printf("%lld steps in %.3f s: %.3f ms/step, %.3e interactions/s; replicas identical: %s\n", (long long)steps, wall,
                            1e3 * wall / (double)steps, inter / wall, identical ? "yes" : "NO");
            for (int i = 0; i < (int)inf[5]; ++i) {  // where every local rank's step went (library event totals, ms per step)
                const auto t = ms.readRankTiming(i);
                const double k = (double)std::max<std::int64_t>(1, t.steps);
                std::printf("  rank %d: force kernels %.3f  behind them %.3f  position exchange %.3f on the wire, %.3f as the waiting "
                            "launch saw it  column sums %.3f  host enqueue %.3f  layout refreshes %lld (%.3f ms each)\n", rank + i,
                            t.forceMs / k, t.updateMs / k, t.posExchangeCommMs / k, t.posExchangeWaitMs / k, t.columnSumExchangeMs / k,
                            t.hostEnqueueMs / k, (long long)t.reorders, t.reorders ? t.reorderMs / (double)t.reorders : 0.0);
            }
            if (!final_path.empty()) {
                ms.download(b.pos.data(), b.vel.data());
                if (report)
                    save(final_path, step0 + steps, time0 + (double)steps * dt);
            }
            if (per_process && report)
                std::remove(id_file.c_str());  // every rank has joined the communicator long ago
            return 0;
        }
        nbody::System sys;
        if (auto_mode) sys.initializeAuto(b.n(), device);   // initialize(numBodies), the force mode chosen by the library
        else if (pair_once) sys.initializeShard(b.n(), 0, b.n(), nbody_pair_once_split_len(b.n()), device);
        else sys.initialize(b.n(), device);             // initialize(numBodies)
        sys.setParticlesPosition(b.pos.data());
        sys.setParticlesVelocity(b.vel.data());
        sys.timing(true);
        if (kdk) sys.setKickDriftKick(true);       // velocity Verlet instead of the reference's kick-drift
        if (pair_once && !auto_mode) sys.setPairOnce(true);  // the pair-once force kernel
        if (auto_mode) std::printf("force mode: %s\n", sys.pairOnce() ? "pair-once" : "one-sided");
        if (particle_eps) {                        // the eps column of the velocity records (kernel.cu:223, 237), unused there
            std::vector<float> eps((size_t)b.n());
            for (std::int64_t i = 0; i < b.n(); ++i) eps[(size_t)i] = b.vel[4 * (size_t)i + 3];
            sys.setParticleSoftening(eps.data());
        }
        nbody::System::Energy e0{};
        if (energy_every > 0) {
            e0 = sys.energy(softening);
            std::printf("step %lld  E = %.9e (K %.6e U %.6e)\n", (long long)step0, e0.total, e0.kinetic, e0.potential);
        }
        const auto t0 = std::chrono::steady_clock::now();
        auto upload = [&]() {  // b in the device's order -> the context (and the eps column as softening lengths)
            sys.setParticlesPosition(b.pos.data());
            sys.setParticlesVelocity(b.vel.data());
            if (particle_eps) {
                std::vector<float> eps((size_t)b.n());
                for (std::int64_t i = 0; i < b.n(); ++i) eps[(size_t)i] = b.vel[4 * (size_t)i + 3];
                sys.setParticleSoftening(eps.data());
            }
        };
        for (std::int64_t s = 1; s <= steps; ++s) {
            if (!order.empty() && reorder_every > 0 && s > 1 && (s - 1) % reorder_every == 0) {
                // a new curve through the current positions (the schedule of nbody_multi_set_reorder_period): the state in the
                // file's order, sorted again, back to the device
                sys.download(b.pos.data(), b.vel.data());
                nbody_io::Bodies given = b;
                for (std::int64_t k = 0; k < b.n(); ++k)
                    for (int c = 0; c < 4; ++c) {
                        given.pos[4 * (size_t)order[(size_t)k] + c] = b.pos[4 * (size_t)k + c];
                        given.vel[4 * (size_t)order[(size_t)k] + c] = b.vel[4 * (size_t)k + c];
                    }
                if (nbody_morton_order(given.pos.data(), b.n(), order.data()) != NBODY_OK) throw std::runtime_error("nbody_morton_order failed");
                for (std::int64_t k = 0; k < b.n(); ++k)
                    for (int c = 0; c < 4; ++c) {
                        b.pos[4 * (size_t)k + c] = given.pos[4 * (size_t)order[(size_t)k] + c];
                        b.vel[4 * (size_t)k + c] = given.vel[4 * (size_t)order[(size_t)k] + c];
                    }
                upload();
            }
            sys.step(dt, softening);                    // the bracket kernel.cu:1225-1242
            const std::int64_t gs = step0 + s;
            if (energy_every > 0 && (s % energy_every == 0 || s == steps)) {
                auto e = sys.energy(softening);
                auto p = sys.momentum();
                std::printf("step %lld  E = %.9e  dE/E0 = %+.3e  |p| = %.3e\n", (long long)gs, e.total, (e.total - e0.total) / std::fabs(e0.total),
                            std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
            }
            if (dump_every > 0 && s % dump_every == 0) {
                sys.download(b.pos.data(), b.vel.data());
                char name[512];
                std::snprintf(name, sizeof name, "%s_%06lld.nbs", dump_prefix.c_str(), (long long)gs);
                save(name, gs, time0 + (double)s * dt);
            }
        }
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto tm = sys.readTiming();
        const double inter = (double)b.n() * (double)b.n() * (double)steps;
        std::printf("%lld steps in %.3f s: %.3f ms/step, %.3e interactions/s (force kernels %.3f ms/step, update %.3f ms/step)\n",
                    (long long)steps, wall, 1e3 * wall / (double)steps, inter / wall, tm.forceMs / (double)steps, tm.updateMs / (double)steps);
        if (!final_path.empty()) {
            sys.download(b.pos.data(), b.vel.data());
            save(final_path, step0 + steps, time0 + (double)steps * dt);
        }
    } catch (const std::exception &ex) {
        std::cerr << "nbody_run: " << ex.what() << "\n";
        return 1;
    }
    return 0;
}
