// nbody_io.hpp -- host-side input/output of the C++ host: the reference's dataset formats
// (main_project/kernel.cu:190-556, component C3 of SURVEY.md), a seeded Plummer sphere, and raw snapshots.
// Mirrors n_body_problem_amd/datasets.py and initial_conditions.py (same formats, same quirk fixes).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace nbody_io {

struct Bodies {
    std::vector<float> pos;  // n x {x,y,z,mass}
    std::vector<float> vel;  // n x {vx,vy,vz,eps}
    std::int64_t n() const { return (std::int64_t)pos.size() / 4; }
    void push(float x, float y, float z, float m, float vx, float vy, float vz, float eps)
    {
        pos.insert(pos.end(), {x, y, z, m});
        vel.insert(vel.end(), {vx, vy, vz, eps});
    }
};

// ---- Tipsy binary, kernel.cu:103-128 (structs) and :190-282 -------------------------------------------
#pragma pack(push, 1)
struct TipsyHeader { double time; std::int32_t nbodies, ndim, nsph, ndark, nstar, pad; };
struct TipsyDark { float mass, pos[3], vel[3], eps; std::int32_t phi; };
struct TipsyStar { float mass, pos[3], vel[3], metals, tform, eps; std::int32_t phi; };
#pragma pack(pop)
static_assert(sizeof(TipsyHeader) == 32 && sizeof(TipsyDark) == 36 && sizeof(TipsyStar) == 44, "Tipsy record sizes");

inline Bodies read_tipsy(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    TipsyHeader h;
    in.read(reinterpret_cast<char *>(&h), sizeof h);
    Bodies b;
    for (int i = 0; i < h.nbodies && in; ++i) {
        if (i < h.ndark) {
            TipsyDark d;
            in.read(reinterpret_cast<char *>(&d), sizeof d);
            b.push(d.pos[0], d.pos[1], d.pos[2], d.mass, d.vel[0], d.vel[1], d.vel[2], d.eps);
        } else {
            TipsyStar s;
            in.read(reinterpret_cast<char *>(&s), sizeof s);
            b.push(s.pos[0], s.pos[1], s.pos[2], s.mass, s.vel[0], s.vel[1], s.vel[2], s.eps);
        }
    }
    if (!in || b.n() != h.nbodies) throw std::runtime_error("truncated Tipsy file " + path);
    return b;
}

inline std::vector<double> read_numbers(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<double> v;
    double x;
    while (in >> x) v.push_back(x);
    return v;
}

// "mass x y z vx vy vz" per body, kernel.cu:321-323 (blank trailing line does not become a body)
inline Bodies read_tab(const std::string &path)
{
    auto t = read_numbers(path);
    if (t.size() % 7) throw std::runtime_error(path + ": not 7 numbers per body");
    Bodies b;
    for (size_t i = 0; i < t.size(); i += 7)
        b.push((float)t[i + 1], (float)t[i + 2], (float)t[i + 3], (float)t[i], (float)t[i + 4], (float)t[i + 5], (float)t[i + 6], 0.f);
    return b;
}

// "z y x vz vy vx" per body, mass 1, kernel.cu:379-387; a token stream, so records wrapped over two lines survive
inline Bodies read_dat(const std::string &path)
{
    auto t = read_numbers(path);
    if (t.size() % 6) throw std::runtime_error(path + ": not 6 numbers per body");
    Bodies b;
    for (size_t i = 0; i < t.size(); i += 6)
        b.push((float)t[i + 2], (float)t[i + 1], (float)t[i], 1.0f, (float)t[i + 5], (float)t[i + 4], (float)t[i + 3], 0.f);
    return b;
}

// n, ndim, time, masses, positions, velocities, eps: kernel.cu:445-529
inline Bodies read_snap(const std::string &path)
{
    auto t = read_numbers(path);
    if (t.size() < 3) throw std::runtime_error(path + ": empty snap file");
    const std::int64_t n = (std::int64_t)t[0];
    if ((int)t[1] != 3 || (std::int64_t)t.size() < 3 + 8 * n) throw std::runtime_error(path + ": not a 3-d snap file");
    const double *m = &t[3], *x = m + n, *v = x + 3 * n, *e = v + 3 * n;
    Bodies b;
    for (std::int64_t i = 0; i < n; ++i)
        b.push((float)x[3 * i], (float)x[3 * i + 1], (float)x[3 * i + 2], (float)m[i], (float)v[3 * i], (float)v[3 * i + 1],
               (float)v[3 * i + 2], (float)e[i]);
    return b;
}

// ---- raw snapshots (n_body_problem_amd/datasets.py: SNAP_HEADER "<8sIIqqd") -----------------------------
#pragma pack(push, 1)
struct SnapHeader { char magic[8]; std::uint32_t version, reserved; std::int64_t n, step; double time; };
#pragma pack(pop)
static_assert(sizeof(SnapHeader) == 40, "snapshot header");

inline void save_snapshot(const std::string &path, const Bodies &b, std::int64_t step, double time)
{
    const std::string tmp = path + ".tmp";
    {
        std::ofstream out(tmp, std::ios::binary);
        if (!out) throw std::runtime_error("cannot write " + tmp);
        SnapHeader h{};
        std::memcpy(h.magic, "NBODYAMD", 8);
        h.version = 1;
        h.n = b.n();
        h.step = step;
        h.time = time;
        out.write(reinterpret_cast<const char *>(&h), sizeof h);
        out.write(reinterpret_cast<const char *>(b.pos.data()), (std::streamsize)(b.pos.size() * sizeof(float)));
        out.write(reinterpret_cast<const char *>(b.vel.data()), (std::streamsize)(b.vel.size() * sizeof(float)));
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp);
}

inline Bodies load_snapshot(const std::string &path, std::int64_t *step = nullptr, double *time = nullptr)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    SnapHeader h;
    in.read(reinterpret_cast<char *>(&h), sizeof h);
    if (!in || std::memcmp(h.magic, "NBODYAMD", 8) != 0 || h.version != 1) throw std::runtime_error(path + ": not an nbody snapshot");
    Bodies b;
    b.pos.resize((size_t)h.n * 4);
    b.vel.resize((size_t)h.n * 4);
    in.read(reinterpret_cast<char *>(b.pos.data()), (std::streamsize)(b.pos.size() * sizeof(float)));
    in.read(reinterpret_cast<char *>(b.vel.data()), (std::streamsize)(b.vel.size() * sizeof(float)));
    if (!in) throw std::runtime_error(path + ": truncated snapshot");
    if (step) *step = h.step;
    if (time) *time = h.time;
    return b;
}

inline Bodies read_any(const std::string &path)
{
    auto ends = [&](const char *e) { size_t k = std::strlen(e); return path.size() >= k && path.compare(path.size() - k, k, e) == 0; };
    if (ends(".bin") || ends(".tipsy")) return read_tipsy(path);
    if (ends(".tab")) return read_tab(path);
    if (ends(".dat")) return read_dat(path);
    if (ends(".snap")) return read_snap(path);
    if (ends(".nbs")) return load_snapshot(path);
    throw std::runtime_error("unknown dataset extension: " + path);
}

// load_data(choice), kernel.cu:975-1013 (ids 4/5 get the .snap parser they should have had)
inline const char *reference_dataset(int choice)
{
    static const char *names[] = {"galaxy_20K.bin", "dubinski.tab", "tab65536.tab", "stars.dat", "k17c.snap", "k17hp.snap"};
    return choice >= 0 && choice <= 5 ? names[choice] : nullptr;
}

// the reference's zero-mass padding to roundup(n,256)+1, kernel.cu:260-278 (optional here)
inline void pad_reference_style(Bodies &b)
{
    const std::int64_t n = b.n(), npad = (n + 255) / 256 * 256 + 1;
    b.pos.resize((size_t)npad * 4, 0.f);
    b.vel.resize((size_t)npad * 4, 0.f);
}

// ---- seeded Plummer sphere (the algorithm of initial_conditions.plummer) ------------------------------
inline std::uint64_t splitmix64(std::uint64_t seed, std::uint64_t counter)
{
    std::uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double uniform01(std::uint64_t seed, std::uint64_t body, unsigned draw, unsigned attempt = 0)
{
    const std::uint64_t ctr = (body << 16) | ((std::uint64_t)(draw & 0xFF) << 8) | (attempt & 0xFF);
    return ((double)(splitmix64(seed, ctr) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
inline Bodies plummer(std::int64_t n, std::uint64_t seed, double r_max = 10.0)
{
    const double pi = 3.14159265358979323846, x1max = std::pow(1.0 + 1.0 / (r_max * r_max), -1.5);
    std::vector<double> p((size_t)n * 3), v((size_t)n * 3);
    double cp[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
    for (std::int64_t i = 0; i < n; ++i) {
        const double r = 1.0 / std::sqrt(std::pow(uniform01(seed, i, 0) * x1max, -2.0 / 3.0) - 1.0);
        double cz = 1.0 - 2.0 * uniform01(seed, i, 1), ph = 2.0 * pi * uniform01(seed, i, 2);
        double sxy = std::sqrt(std::fmax(0.0, 1.0 - cz * cz));
        p[3 * i] = r * sxy * std::cos(ph);
        p[3 * i + 1] = r * sxy * std::sin(ph);
        p[3 * i + 2] = r * cz;
        double q = 0.0;
        for (unsigned a = 0; a < 256; ++a) {
            const double x4 = uniform01(seed, i, 3, a), x5 = uniform01(seed, i, 4, a);
            if (0.1 * x5 < x4 * x4 * std::pow(1.0 - x4 * x4, 3.5)) { q = x4; break; }
        }
        const double speed = q * std::sqrt(2.0) * std::pow(1.0 + r * r, -0.25);
        cz = 1.0 - 2.0 * uniform01(seed, i, 5);
        ph = 2.0 * pi * uniform01(seed, i, 6);
        sxy = std::sqrt(std::fmax(0.0, 1.0 - cz * cz));
        v[3 * i] = speed * sxy * std::cos(ph);
        v[3 * i + 1] = speed * sxy * std::sin(ph);
        v[3 * i + 2] = speed * cz;
        for (int c = 0; c < 3; ++c) { cp[c] += p[3 * i + c]; cv[c] += v[3 * i + c]; }
    }
    Bodies b;
    b.pos.reserve((size_t)n * 4);
    b.vel.reserve((size_t)n * 4);
    for (std::int64_t i = 0; i < n; ++i)
        b.push((float)(p[3 * i] - cp[0] / n), (float)(p[3 * i + 1] - cp[1] / n), (float)(p[3 * i + 2] - cp[2] / n), (float)(1.0 / n),
               (float)(v[3 * i] - cv[0] / n), (float)(v[3 * i + 1] - cv[1] / n), (float)(v[3 * i + 2] - cv[2] / n), 0.f);
    return b;
}

}  // namespace nbody_io
