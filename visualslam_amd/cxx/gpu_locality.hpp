// Host-side placement of one rank next to its GPU ("the curve mainly exposes host-side feeding").
// With eight host-fed camera streams on one node every rank moves ~1 GB per batch over its GPU's PCIe link; staging buffers
// and worker threads on the far socket put that traffic on the inter-socket fabric as well.  A rank therefore
//   1. asks HIP for its GPU's PCI address (hipDeviceGetPCIBusId),
//   2. reads the NUMA node of that address from sysfs (/sys/bus/pci/devices/<bdf>/numa_node),
//   3. restricts its own thread - and with it every thread it creates later - to that node's CPUs
//      (/sys/devices/system/node/node<k>/cpulist, intersected with the CPUs the process may use at all), and
//   4. only then allocates its pinned staging buffers: the kernel's default first-touch policy places them on the node
//      the allocating thread runs on.
// Header-only, plain Linux calls, no libnuma.  Everything is best effort: a missing file, node -1 (no affinity known) or an
// empty intersection leaves the thread where it was, and `Placement` says what happened.
#pragma once
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace vslam {
namespace locality {

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; malformed pieces are skipped
inline std::vector<int> parse_cpulist(const std::string& text) {
    std::vector<int> cpus;
    std::stringstream ss(text);
    std::string piece;
    while (std::getline(ss, piece, ',')) {
        while (!piece.empty() && std::isspace((unsigned char)piece.back())) piece.pop_back();
        while (!piece.empty() && std::isspace((unsigned char)piece.front())) piece.erase(piece.begin());
        if (piece.empty()) continue;
        int a = 0, b = 0;
        if (std::sscanf(piece.c_str(), "%d-%d", &a, &b) == 2) {
            for (int c = a; c <= b && b - a < (1 << 16); ++c) cpus.push_back(c);
        } else if (std::sscanf(piece.c_str(), "%d", &a) == 1) {
            cpus.push_back(a);
        }
    }
    std::sort(cpus.begin(), cpus.end());
    cpus.erase(std::unique(cpus.begin(), cpus.end()), cpus.end());
    return cpus;
}

// {0,1,2,3,8} -> "0-3,8"
inline std::string format_cpulist(const std::vector<int>& cpus) {
    std::string out;
    for (size_t i = 0; i < cpus.size();) {
        size_t j = i;
        while (j + 1 < cpus.size() && cpus[j + 1] == cpus[j] + 1) ++j;
        if (!out.empty()) out += ",";
        out += std::to_string(cpus[i]);
        if (j > i) out += "-" + std::to_string(cpus[j]);
        i = j + 1;
    }
    return out;
}

inline bool read_first_line(const std::string& path, std::string* out) {
    std::ifstream f(path);
    if (!f) return false;
    std::getline(f, *out);
    return true;
}

// HIP prints the address as "0000:C1:00.0"; sysfs names the directory in lower case
inline std::string normalise_bdf(std::string bdf) {
    for (char& c : bdf) c = (char)std::tolower((unsigned char)c);
    while (!bdf.empty() && (bdf.back() == '\0' || std::isspace((unsigned char)bdf.back()))) bdf.pop_back();
    return bdf;
}

// NUMA node of a PCI device, -1 when sysfs does not know (single-node machines report -1)
inline int numa_node_of_pci(const std::string& bdf, const std::string& sysfs_root = "/sys") {
    std::string line;
    if (!read_first_line(sysfs_root + "/bus/pci/devices/" + normalise_bdf(bdf) + "/numa_node", &line)) return -1;
    int node = -1;
    if (std::sscanf(line.c_str(), "%d", &node) != 1) return -1;
    return node;
}

inline std::vector<int> cpus_of_node(int node, const std::string& sysfs_root = "/sys") {
    std::string line;
    if (node < 0 || !read_first_line(sysfs_root + "/devices/system/node/node" + std::to_string(node) + "/cpulist", &line)) return {};
    return parse_cpulist(line);
}

// CPUs the calling thread may run on right now (affinity mask: a container's cpuset shows up here)
inline std::vector<int> current_affinity() {
    std::vector<int> cpus;
    cpu_set_t* set = CPU_ALLOC(4096);
    if (!set) return cpus;
    const size_t sz = CPU_ALLOC_SIZE(4096);
    CPU_ZERO_S(sz, set);
    if (sched_getaffinity(0, sz, set) == 0)
        for (int c = 0; c < 4096; ++c)
            if (CPU_ISSET_S(c, sz, set)) cpus.push_back(c);
    CPU_FREE(set);
    return cpus;
}

inline std::vector<int> intersect(const std::vector<int>& a, const std::vector<int>& b) {
    std::vector<int> out;
    std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(out));
    return out;
}

inline bool bind_current_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return false;
    cpu_set_t* set = CPU_ALLOC(4096);
    if (!set) return false;
    const size_t sz = CPU_ALLOC_SIZE(4096);
    CPU_ZERO_S(sz, set);
    for (int c : cpus)
        if (c >= 0 && c < 4096) CPU_SET_S(c, sz, set);
    const bool ok = sched_setaffinity(0, sz, set) == 0;
    CPU_FREE(set);
    return ok;
}

struct Placement {
    int rank = 0, gpu = 0, numa_node = -1;
    std::string pci_bus_id;
    std::vector<int> node_cpus;  // the node's CPUs as sysfs lists them
    std::vector<int> cpus;       // what this rank's threads run on afterwards
    bool bound = false;          // the affinity of the calling thread was narrowed to `cpus`
    std::string note;

    std::string json() const {
        char head[256];
        std::snprintf(head, sizeof(head), "{\"rank\": %d, \"gpu\": %d, \"pci_bus_id\": \"%s\", \"numa_node\": %d, \"bound\": %s, ", rank, gpu,
                      pci_bus_id.c_str(), numa_node, bound ? "true" : "false");
        return std::string(head) + "\"cpus\": \"" + format_cpulist(cpus) + "\", \"n_cpus\": " + std::to_string(cpus.size()) + ", \"note\": \"" + note + "\"}";
    }
};

// Steps 2-3 of the header comment for a GPU whose PCI address is known; `bind` = false only reports.
inline Placement place_near_pci(int rank, int gpu, const std::string& bdf, bool bind = true, const std::string& sysfs_root = "/sys") {
    Placement p;
    p.rank = rank, p.gpu = gpu, p.pci_bus_id = normalise_bdf(bdf);
    const std::vector<int> allowed = current_affinity();
    p.cpus = allowed;
    p.numa_node = numa_node_of_pci(bdf, sysfs_root);
    if (p.numa_node < 0) {
        p.note = "sysfs reports no NUMA node for this device: threads stay where they are";
        return p;
    }
    p.node_cpus = cpus_of_node(p.numa_node, sysfs_root);
    const std::vector<int> want = intersect(p.node_cpus, allowed);
    if (want.empty()) {
        p.note = p.node_cpus.empty() ? "the node's cpulist is missing" : "none of the node's CPUs is in this process's affinity mask: threads stay where they are";
        return p;
    }
    if (!bind) {
        p.cpus = want;
        p.note = "report only";
        return p;
    }
    if (bind_current_thread(want)) {
        p.cpus = want;
        p.bound = true;
        p.note = "threads and first-touch allocations of this rank stay on the GPU's node";
    } else {
        p.note = "sched_setaffinity failed: threads stay where they are";
    }
    return p;
}

}  // namespace locality
}  // namespace vslam
