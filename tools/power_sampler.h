// tools/power_sampler.h -- samples the GPU's power / clock telemetry from sysfs while a kernel loop runs (the
// power cap measured, not inferred).  Sources, whichever exist on the box: hwmon power1_average / power1_input (uW), power1_cap (uW),
// freq1_input (Hz, sclk), the starred level of pp_dpm_sclk (MHz), and the gpu_metrics blob's header (format revision, printed so
// that its fields can be decoded offline).  One sample every `period_ms`; a leg = start() .. stop() -> summary line + decimated trace.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>

struct PowerSampler {
    struct Src { std::string name, path; int kind; };   // kind 0: plain integer file, 1: pp_dpm (starred line)
    std::vector<Src> srcs;
    std::vector<std::vector<double>> rows;   // [sample][0 = t_ms, 1.. = sources]
    std::atomic<bool> run{false};
    std::thread th;
    int period_ms = 10;

    static bool read_text(const std::string& p, char* buf, size_t n) {
        FILE* f = fopen(p.c_str(), "r");
        if (!f) return false;
        const size_t k = fread(buf, 1, n - 1, f);
        fclose(f);
        buf[k] = 0;
        return k > 0;
    }
    static std::vector<std::string> ls(const std::string& d) {
        std::vector<std::string> out;
        if (DIR* dir = opendir(d.c_str())) {
            while (dirent* e = readdir(dir)) if (e->d_name[0] != '.') out.push_back(e->d_name);
            closedir(dir);
        }
        return out;
    }
    // every card's hwmon directory; verbose: print what is there (first call of a session: learn the box)
    void discover(bool verbose) {
        char buf[4096];
        for (const std::string& c : ls("/sys/class/drm")) {
            if (c.compare(0, 4, "card") != 0 || c.find('-') != std::string::npos) continue;
            const std::string dev = "/sys/class/drm/" + c + "/device";
            for (const std::string& h : ls(dev + "/hwmon")) {
                const std::string hd = dev + "/hwmon/" + h;
                if (verbose) printf("# %s:", hd.c_str());
                for (const std::string& f : ls(hd)) {
                    if (verbose) printf(" %s", f.c_str());
                    const bool want = f == "power1_average" || f == "power1_input" || f == "power1_cap" || f == "freq1_input" || f == "freq2_input" ||
                                      f == "temp1_input" || f == "temp2_input" || f == "in0_input";
                    if (want && read_text(hd + "/" + f, buf, sizeof buf)) srcs.push_back({c + "." + f, hd + "/" + f, 0});
                }
                if (verbose) printf("\n");
            }
            for (const char* f : {"pp_dpm_sclk", "pp_dpm_mclk"})
                if (read_text(dev + "/" + f, buf, sizeof buf)) {
                    srcs.push_back({c + "." + f, dev + "/" + f, 1});
                    if (verbose) { for (char* q = buf; *q; ++q) if (*q == '\n') *q = '|'; printf("# %s/%s: %s\n", dev.c_str(), f, buf); }
                }
            if (verbose) {
                for (const char* f : {"power_dpm_force_performance_level", "pp_power_profile_mode", "gpu_busy_percent"})
                    if (read_text(dev + "/" + f, buf, sizeof buf)) { for (char* q = buf; *q; ++q) if (*q == '\n') *q = '|'; printf("# %s/%s: %.300s\n", dev.c_str(), f, buf); }
                FILE* g = fopen((dev + "/gpu_metrics").c_str(), "rb");
                if (g) {
                    unsigned char hb[4096]; const size_t k = fread(hb, 1, sizeof hb, g); fclose(g);
                    if (k >= 4) printf("# %s/gpu_metrics: %zu bytes, structure_size %u, format_revision %u, content_revision %u\n", dev.c_str(), k, hb[0] | (hb[1] << 8), hb[2], hb[3]);
                }
            }
        }
        if (verbose) { printf("# sampled:"); for (auto& s : srcs) printf(" %s", s.name.c_str()); printf("\n"); }
    }
    double read_src(const Src& s) {
        char buf[1024];
        if (!read_text(s.path, buf, sizeof buf)) return -1;
        if (s.kind == 0) return atof(buf);
        for (char* l = strtok(buf, "\n"); l; l = strtok(nullptr, "\n"))
            if (strchr(l, '*')) { const char* c = strchr(l, ':'); return c ? atof(c + 1) : -1; }
        return -1;
    }
    void start() {
        rows.clear();
        run = true;
        th = std::thread([this] {
            const auto t0 = std::chrono::steady_clock::now();
            while (run) {
                std::vector<double> r;
                r.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                for (auto& s : srcs) r.push_back(read_src(s));
                rows.push_back(std::move(r));
                std::this_thread::sleep_for(std::chrono::milliseconds(period_ms));
            }
        });
    }
    // summary over the samples with t >= skip_ms, then every `every`-th row
    void stop(const char* leg, double skip_ms, int every) {
        run = false;
        th.join();
        printf("leg %s: %zu samples over %.0f ms", leg, rows.size(), rows.empty() ? 0.0 : rows.back()[0]);
        for (size_t c = 0; c < srcs.size(); ++c) {
            double s = 0, mn = 1e300, mx = -1e300; int n = 0;
            for (auto& r : rows) if (r[0] >= skip_ms) { const double v = r[c + 1]; s += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; ++n; }
            if (n) printf(" | %s mean %.6g min %.6g max %.6g", srcs[c].name.c_str(), s / n, mn, mx);
        }
        printf("\n");
        if (every > 0) {
            printf("  trace (t_ms");
            for (auto& s : srcs) printf(", %s", s.name.c_str());
            printf("):\n");
            for (size_t i = 0; i < rows.size(); i += every) {
                printf("   %8.1f", rows[i][0]);
                for (size_t c = 1; c < rows[i].size(); ++c) printf(" %12.6g", rows[i][c]);
                printf("\n");
            }
        }
    }
};
