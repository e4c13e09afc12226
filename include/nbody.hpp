// nbody.hpp -- header-only C++ convenience layer over the C ABI of nbody.h.
//
// Keeps the names of the reference's host interface for this path (main_project/kernel.cu):
//   initialize(numBodies)            :130-161      -> nbody::System::initialize / constructor
//   setParticlesPosition(real*)      :163-177      -> System::setParticlesPosition
//   setParticlesVelocity(real*)      :179-188      -> System::setParticlesVelocity
//   the per-frame bracket            :1225-1242    -> System::step(dt, softening)
// so that the reference's main loop reads the same after the swap.  Errors become
// std::runtime_error carrying the library's message; the C ABI itself never throws.
#pragma once
#include "nbody.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace nbody {

constexpr float kTimeTick = 0.008f;            // TIME_TICK, kernel.cu:63
constexpr float kSofteningVersion3 = 1.0e-2f;  // effective eps of cal_single_acclerate_without_mass_new, :665-692
constexpr float kSofteningVersion1 = 1.0e-3f;  // sqrt(EPSILON) of cal_single_acclerate, :808-824

class System {
public:
    System() = default;
    explicit System(std::int64_t numBodies, int device = 0) { initialize(numBodies, device); }
    System(const System &) = delete;
    System &operator=(const System &) = delete;
    ~System() { nbody_destroy(ctx_); }

    void initialize(std::int64_t numBodies, int device = 0)
    {
        nbody_destroy(ctx_);
        ctx_ = nullptr;
        check(nbody_create(&ctx_, device, numBodies), "nbody_create");
        n_ = numBodies;
    }
    // initialize(numBodies) with the faster force mode for this body count already selected (nbody_create_auto): the
    // pair-once kernels from NBODY_PAIR_ONCE_MIN_BODIES bodies on (round 4: at every size), the one-sided ones below.
    void initializeAuto(std::int64_t numBodies, int device = 0)
    {
        nbody_destroy(ctx_);
        ctx_ = nullptr;
        check(nbody_create_auto(&ctx_, device, numBodies), "nbody_create_auto");
        n_ = numBodies;
    }
    bool pairOnce() const { return nbody_force_mode(ctx_) == NBODY_FORCE_SYMMETRIC; }
    // One rank of a sharded run: rows [rowLo, rowLo+rowCount) against all numBodies columns.
    void initializeShard(std::int64_t numBodies, std::int64_t rowLo, std::int64_t rowCount, std::int64_t splitLen = 0,
                         int device = 0)
    {
        nbody_destroy(ctx_);
        ctx_ = nullptr;
        check(nbody_create_shard(&ctx_, device, numBodies, rowLo, rowCount, splitLen), "nbody_create_shard");
        n_ = numBodies;
    }

    void setParticlesPosition(const float *xyzm) { check(nbody_set_positions(ctx_, xyzm), "nbody_set_positions"); }
    void setParticlesVelocity(const float *xyzw) { check(nbody_set_velocities(ctx_, xyzw), "nbody_set_velocities"); }
    void download(float *xyzm, float *xyzw) { check(nbody_download(ctx_, xyzm, xyzw), "nbody_download"); }

    // The mapped-pointer analogue of cudaGraphicsResourceGetMappedPointer (kernel.cu:1226).
    float *positionsDevice() { return nbody_positions_device(ctx_); }
    float *velocitiesDevice() { return nbody_velocities_device(ctx_); }

    // One synchronous step on the owned buffers (the reference synchronises after each kernel, :1232,1236).
    void step(float dt = kTimeTick, float softening = kSofteningVersion3)
    {
        check(nbody_step(ctx_, positionsDevice(), velocitiesDevice(), nullptr, dt, softening), "nbody_step");
    }
    // step(positions, velocities, masses, dt, softening) on caller-owned device buffers.
    void step(float *dPositions, float *dVelocities, const float *dMasses, float dt, float softening)
    {
        check(nbody_step(ctx_, dPositions, dVelocities, dMasses, dt, softening), "nbody_step");
    }
    void stepN(int k, float dt, float softening) { check(nbody_step_n(ctx_, k, dt, softening), "nbody_step_n"); }
    // false: the reference's kick-drift (kernel.cu:777-801); true: velocity Verlet with cached accelerations
    void setKickDriftKick(bool on)
    {
        check(nbody_set_integrator(ctx_, on ? NBODY_INTEGRATOR_KDK : NBODY_INTEGRATOR_KICK_DRIFT), "nbody_set_integrator");
    }
    // experimental pair-once force kernel (context created with splitLen = nbody_pair_once_split_len(numBodies))
    void setPairOnce(bool on)
    {
        check(nbody_set_force_mode(ctx_, on ? NBODY_FORCE_SYMMETRIC : NBODY_FORCE_ONE_SIDED), "nbody_set_force_mode");
    }

    // per-particle softening lengths of all numBodies bodies (host floats, e.g. velocities[4i+3] of the reference's
    // loaders, kernel.cu:223); nullptr switches it off.  eps_ij^2 = softening^2 + eps_i^2 + eps_j^2.
    void setParticleSoftening(const float *hostEps)
    {
        check(nbody_upload_particle_softening(ctx_, hostEps), "nbody_upload_particle_softening");
    }

    struct Energy { double kinetic, potential, total; };
    Energy energy(float softening)
    {
        double e[3];
        check(nbody_energy(ctx_, positionsDevice(), velocitiesDevice(), softening, e), "nbody_energy");
        return {e[0], e[1], e[2]};
    }
    std::vector<double> momentum()
    {
        std::vector<double> p(4);
        check(nbody_momentum(ctx_, positionsDevice(), velocitiesDevice(), p.data()), "nbody_momentum");
        return p;
    }

    void timing(bool on) { check(nbody_timing_enable(ctx_, on ? 1 : 0), "nbody_timing_enable"); }
    struct Timing { double forceMs, updateMs; std::int64_t forceLaunches, updateLaunches; };
    Timing readTiming()
    {
        Timing t{};
        check(nbody_timing_read(ctx_, &t.forceMs, &t.forceLaunches, &t.updateMs, &t.updateLaunches), "nbody_timing_read");
        return t;
    }

    std::int64_t numBodies() const { return n_; }
    nbody_ctx *handle() { return ctx_; }

private:
    void check(int status, const char *what)
    {
        if (status != NBODY_OK)
            throw std::runtime_error(std::string(what) + ": " + nbody_last_error(ctx_) + " (" +
                                     nbody_status_string(status) + ")");
    }
    nbody_ctx *ctx_ = nullptr;
    std::int64_t n_ = 0;
};

// The same host interface for a body set whose rows are sharded over several GPUs of one node, driven from this one
// host thread (nbody_multi_create: one RCCL communicator per device, the exchange inside the library).
class MultiSystem {
public:
    MultiSystem() = default;
    MultiSystem(const MultiSystem &) = delete;
    MultiSystem &operator=(const MultiSystem &) = delete;
    ~MultiSystem() { nbody_multi_destroy(m_); }

    // initialize(numBodies) on the given devices; pairOnce / kickDriftKick / ring / peerCopy select the variants
    void initialize(std::int64_t numBodies, const std::vector<int> &devices, bool pairOnce = false, bool kickDriftKick = false,
                    bool ring = false, bool peerCopy = false, std::int64_t splitLen = 0, bool mortonOrder = false,
                    bool autoMode = false)
    {
        nbody_multi_destroy(m_);
        m_ = nullptr;
        nbody_multi_config cfg{};
        cfg.n_bodies = numBodies;
        cfg.split_len = splitLen;
        cfg.force_mode = autoMode ? NBODY_FORCE_AUTO : pairOnce ? NBODY_FORCE_SYMMETRIC : NBODY_FORCE_ONE_SIDED;
        cfg.integrator = kickDriftKick ? NBODY_INTEGRATOR_KDK : NBODY_INTEGRATOR_KICK_DRIFT;
        cfg.exchange = ring ? NBODY_EXCHANGE_RING : NBODY_EXCHANGE_ALLGATHER;
        cfg.transport = peerCopy ? NBODY_TRANSPORT_PEER_COPY : NBODY_TRANSPORT_RCCL;
        cfg.body_order = mortonOrder ? NBODY_ORDER_MORTON : NBODY_ORDER_GIVEN;  // stored along a Morton curve, downloads undo it
        check(nbody_multi_create(&m_, &cfg, devices.data(), (int)devices.size()), "nbody_multi_create");
        n_ = numBodies;
    }
    // One rank per process (mpirun / torchrun --no-python / a job script): rank 0 calls uniqueId() and hands the bytes to the
    // other ranks by any channel; every rank then calls initializeRank with them.  From setState on all calls are collective.
    static std::vector<unsigned char> uniqueId()
    {
        std::vector<unsigned char> id(NBODY_UNIQUE_ID_BYTES);
        if (nbody_multi_unique_id(id.data()) != NBODY_OK)
            throw std::runtime_error(std::string("nbody_multi_unique_id: ") + nbody_multi_last_error(nullptr));
        return id;
    }
    void initializeRank(std::int64_t numBodies, int device, int rank, int worldSize, const std::vector<unsigned char> &id,
                        bool pairOnce = false, bool kickDriftKick = false, bool ring = false, std::int64_t splitLen = 0,
                        bool mortonOrder = false, bool autoMode = false)
    {
        if (id.size() != NBODY_UNIQUE_ID_BYTES)
            throw std::runtime_error("MultiSystem::initializeRank: the unique id has NBODY_UNIQUE_ID_BYTES bytes");
        nbody_multi_destroy(m_);
        m_ = nullptr;
        nbody_multi_config cfg{};
        cfg.n_bodies = numBodies;
        cfg.split_len = splitLen;
        cfg.force_mode = autoMode ? NBODY_FORCE_AUTO : pairOnce ? NBODY_FORCE_SYMMETRIC : NBODY_FORCE_ONE_SIDED;
        cfg.integrator = kickDriftKick ? NBODY_INTEGRATOR_KDK : NBODY_INTEGRATOR_KICK_DRIFT;
        cfg.exchange = ring ? NBODY_EXCHANGE_RING : NBODY_EXCHANGE_ALLGATHER;
        cfg.transport = NBODY_TRANSPORT_RCCL;
        cfg.body_order = mortonOrder ? NBODY_ORDER_MORTON : NBODY_ORDER_GIVEN;
        check(nbody_multi_create_rank(&m_, &cfg, device, rank, worldSize, id.data()), "nbody_multi_create_rank");
        n_ = numBodies;
    }
    void setState(const float *xyzm, const float *xyzw) { check(nbody_multi_set_state(m_, xyzm, xyzw), "nbody_multi_set_state"); }
    // the reference's two setters, independent copies (kernel.cu:163-188)
    void setParticlesPosition(const float *xyzm) { check(nbody_multi_set_positions(m_, xyzm), "nbody_multi_set_positions"); }
    void setParticlesVelocity(const float *xyzw) { check(nbody_multi_set_velocities(m_, xyzw), "nbody_multi_set_velocities"); }
    void setParticleSoftening(const float *hostEps)
    {
        check(nbody_multi_set_particle_softening(m_, hostEps), "nbody_multi_set_particle_softening");
    }
    void download(float *xyzm, float *xyzw) { check(nbody_multi_download(m_, xyzm, xyzw), "nbody_multi_download"); }
    void reorder() { check(nbody_multi_reorder(m_), "nbody_multi_reorder"); }  // mortonOrder: a new curve through the current positions
    void setReorderPeriod(std::int64_t steps) { check(nbody_multi_set_reorder_period(m_, steps), "nbody_multi_set_reorder_period"); }
    void step(float dt = kTimeTick, float softening = kSofteningVersion3) { check(nbody_multi_step(m_, dt, softening), "nbody_multi_step"); }
    void stepN(int k, float dt, float softening) { check(nbody_multi_step_n(m_, k, dt, softening), "nbody_multi_step_n"); }
    System::Energy energy(float softening)
    {
        double e[3];
        check(nbody_multi_energy(m_, softening, e), "nbody_multi_energy");
        return {e[0], e[1], e[2]};
    }
    std::vector<double> momentum()
    {
        std::vector<double> p(4);
        check(nbody_multi_momentum(m_, p.data()), "nbody_multi_momentum");
        return p;
    }
    bool replicasIdentical()
    {
        std::uint64_t c[2];
        check(nbody_multi_replica_checksums(m_, c), "nbody_multi_replica_checksums");
        return c[0] == c[1];
    }
    // kernels of the shard contexts and the exchanges (nbody_multi_timing_*: HIP events on the streams they run on)
    void timing(bool on) { check(nbody_multi_timing_enable(m_, on ? 1 : 0), "nbody_multi_timing_enable"); }
    // Totals of one local rank since the last read (sums of event-pair durations, ms): where its steps went.
    struct RankTiming {
        std::int64_t steps;
        double hostEnqueueMs, forceMs, updateMs, auxMs, posExchangeCommMs, posExchangeWaitMs, columnSumExchangeMs, reorderMs;
        std::int64_t forceLaunches, updateLaunches, reorders;
    };
    RankTiming readRankTiming(int localIndex)
    {
        double v[16];
        check(nbody_multi_timing_read(m_, localIndex, v), "nbody_multi_timing_read");
        return {(std::int64_t)v[0], v[1], v[2], v[4], v[6], v[8], v[10], v[12], v[14],
                (std::int64_t)v[3], (std::int64_t)v[5], (std::int64_t)v[15]};
    }
    // local rank 0's kernels, in the shape of System::readTiming
    System::Timing readTiming()
    {
        const RankTiming r = readRankTiming(0);
        return {r.forceMs, r.updateMs, r.forceLaunches, r.updateLaunches};
    }
    std::vector<std::int64_t> info()
    {
        std::vector<std::int64_t> v(8);
        check(nbody_multi_info(m_, v.data()), "nbody_multi_info");
        return v;
    }
    std::int64_t numBodies() const { return n_; }
    nbody_multi *handle() { return m_; }

private:
    void check(int status, const char *what)
    {
        if (status != NBODY_OK)
            throw std::runtime_error(std::string(what) + ": " + nbody_multi_last_error(m_) + " (" + nbody_status_string(status) + ")");
    }
    nbody_multi *m_ = nullptr;
    std::int64_t n_ = 0;
};

}  // namespace nbody
