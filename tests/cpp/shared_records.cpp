// shared_records.cpp -- the device-resident hand-off of the particle records
// (psim_shared_aos) between two processes, without any renderer: the consumer stands where a
// Vulkan / wgpu host would, importing the dma-buf descriptor as external memory and reading
// the 64-byte records of src/simulation/mod.rs:86-98 straight from device memory.
//
// The parent forks BEFORE anything touches the GPU.  The producer owns the simulation, steps
// it, exports frame after frame into the shared buffer and tells the consumer -- over a socket
// -- when a frame is complete and what its hash is (computed from an ordinary download); the
// consumer maps the descriptor ONCE and must read every frame, byte for byte, through that one
// mapping.  A build with hipcc is needed only for the HIP runtime's headers and library: there
// is no device code here.
//
// usage: shared_records <particles> <frames>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "psim.hpp"

namespace {

uint64_t hash_bytes(const void *p, size_t bytes)
{
    uint64_t h = 1469598103934665603ull;
    const uint64_t *w = static_cast<const uint64_t *>(p);
    for (size_t i = 0; i < bytes / 8; ++i) h = (h ^ w[i]) * 1099511628211ull;
    return h;
}

int send_fd(int sock, int fd)
{
    char byte = 'f', ctl[CMSG_SPACE(sizeof(int))];
    struct iovec io = {&byte, 1};
    struct msghdr msg;
    std::memset(&msg, 0, sizeof(msg));
    std::memset(ctl, 0, sizeof(ctl));
    msg.msg_iov = &io, msg.msg_iovlen = 1, msg.msg_control = ctl, msg.msg_controllen = sizeof(ctl);
    struct cmsghdr *c = CMSG_FIRSTHDR(&msg);
    c->cmsg_level = SOL_SOCKET, c->cmsg_type = SCM_RIGHTS, c->cmsg_len = CMSG_LEN(sizeof(int));
    std::memcpy(CMSG_DATA(c), &fd, sizeof(int));
    return sendmsg(sock, &msg, 0) == 1 ? 0 : -1;
}

int recv_fd(int sock)
{
    char byte, ctl[CMSG_SPACE(sizeof(int))];
    struct iovec io = {&byte, 1};
    struct msghdr msg;
    std::memset(&msg, 0, sizeof(msg));
    msg.msg_iov = &io, msg.msg_iovlen = 1, msg.msg_control = ctl, msg.msg_controllen = sizeof(ctl);
    if (recvmsg(sock, &msg, 0) != 1) return -1;
    struct cmsghdr *c = CMSG_FIRSTHDR(&msg);
    if (!c || c->cmsg_type != SCM_RIGHTS) return -1;
    int fd;
    std::memcpy(&fd, CMSG_DATA(c), sizeof(int));
    return fd;
}

bool send_all(int sock, const void *p, size_t n) { return write(sock, p, n) == (ssize_t)n; }
bool recv_all(int sock, void *p, size_t n) { return read(sock, p, n) == (ssize_t)n; }

struct FrameNote {
    uint64_t records;      // particles in the frame
    uint64_t hash;         // of records x 64 bytes
};

int producer(int sock, uint32_t n, int frames)
{
    try {
        psim::HipParticleSimulation sim(n, psim::SphereGeneration::Filled);
        const psim::HipParticleSimulation::SharedRecords shared = sim.shared_records();
        if (shared.dmabuf_fd < 0 || shared.bytes < (uint64_t)n * sizeof(psim::Particle)) return 3;
        const uint64_t bytes = shared.bytes;
        if (send_fd(sock, shared.dmabuf_fd) != 0 || !send_all(sock, &bytes, sizeof(bytes))) return 4;
        close(shared.dmabuf_fd);                          // the consumer holds its own reference
        psim::SimParams p = psim::default_sim_params();
        p.is_mouse_dragging = 1u, p.mouse_radius = 30.0f, p.mouse_force = 50.0f, p.color_mode = 1u;
        for (int k = 0; k < frames; ++k) {
            p.mouse_position[0] = -10.0f + 2.0f * static_cast<float>(k);
            sim.update(p);
            sim.export_to(shared.device);                 // behind the update, on the same stream
            sim.sync();                                   // the frame is complete in device memory
            const std::vector<psim::Particle> want = sim.download();
            const FrameNote note = {want.size(), hash_bytes(want.data(), want.size() * sizeof(psim::Particle))};
            char ack = 0;
            if (!send_all(sock, &note, sizeof(note)) || !recv_all(sock, &ack, 1) || ack != 'k') return 5;
        }
    } catch (const psim::Error &e) {
        std::fprintf(stderr, "producer: psim error %d: %s\n", e.code(), e.what());
        return 1;
    }
    return 0;
}

int consumer(int sock, int frames)
{
    const int fd = recv_fd(sock);
    uint64_t bytes = 0;
    if (fd < 0 || !recv_all(sock, &bytes, sizeof(bytes))) return 4;
    hipExternalMemoryHandleDesc desc;
    std::memset(&desc, 0, sizeof(desc));
    desc.type = hipExternalMemoryHandleTypeOpaqueFd;
    desc.handle.fd = fd;
    desc.size = bytes;
    hipExternalMemory_t ext = nullptr;
    hipExternalMemoryBufferDesc range;
    std::memset(&range, 0, sizeof(range));
    range.size = bytes;
    void *mapped = nullptr;
    if (hipImportExternalMemory(&ext, &desc) != hipSuccess ||
        hipExternalMemoryGetMappedBuffer(&mapped, ext, &range) != hipSuccess) {
        std::fprintf(stderr, "consumer: the descriptor could not be imported\n");
        return 6;
    }
    std::vector<char> host(bytes);
    for (int k = 0; k < frames; ++k) {
        FrameNote note;
        if (!recv_all(sock, &note, sizeof(note))) return 5;
        const size_t frame_bytes = note.records * 64u;
        if (frame_bytes > bytes ||
            hipMemcpy(host.data(), mapped, frame_bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return 7;
        if (hash_bytes(host.data(), frame_bytes) != note.hash) {
            std::fprintf(stderr, "consumer: frame %d read through the mapping differs\n", k);
            return 8;
        }
        if (!send_all(sock, "k", 1)) return 5;
    }
    (void)hipDestroyExternalMemory(ext);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc != 3) {
        std::fprintf(stderr, "usage: %s particles frames\n", argv[0]);
        return 2;
    }
    const uint32_t n = static_cast<uint32_t>(std::strtoul(argv[1], nullptr, 10));
    const int frames = std::atoi(argv[2]);
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) return 2;
    const pid_t child = fork();                           // before any GPU call
    if (child < 0) return 2;
    if (child == 0) {
        close(sv[0]);
        _exit(consumer(sv[1], frames));
    }
    close(sv[1]);
    const int rc = producer(sv[0], n, frames);
    close(sv[0]);                                         // a consumer still waiting sees the end
    int status = 0;
    waitpid(child, &status, 0);
    const int crc = WIFEXITED(status) ? WEXITSTATUS(status) : 99;
    std::printf("producer %d consumer %d frames %d particles %u\n", rc, crc, frames, n);
    return rc ? rc : crc;
}
