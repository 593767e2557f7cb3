// exp_dmabuf.hip -- does this runtime hand out a dma-buf for a hipMalloc'ed range, and can a
// second process map it?  (The question behind a device-resident replacement of
// get_particle_buffer: a Vulkan / wgpu host imports dma-buf file descriptors as external memory.)
// The parent forks BEFORE touching the GPU; the producer exports and passes the descriptor over
// a socket (SCM_RIGHTS); the consumer imports it with hipImportExternalMemory and reads it.
//
// build: hipcc --offload-arch=gfx950 -O2 -o exp_dmabuf exp_dmabuf.hip ; run: ./exp_dmabuf
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#define TRY(expr)                                                                   \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));              \
            return 1;                                                               \
        }                                                                           \
    } while (0)

__global__ void fill(uint32_t *p, uint32_t n, uint32_t seed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * 2654435761u + seed;
}

static int send_fd(int sock, int fd)
{
    char byte = 'f', ctl[CMSG_SPACE(sizeof(int))];
    struct iovec io = {&byte, 1};
    struct msghdr msg;
    memset(&msg, 0, sizeof(msg));
    memset(ctl, 0, sizeof(ctl));
    msg.msg_iov = &io, msg.msg_iovlen = 1, msg.msg_control = ctl, msg.msg_controllen = sizeof(ctl);
    struct cmsghdr *c = CMSG_FIRSTHDR(&msg);
    c->cmsg_level = SOL_SOCKET, c->cmsg_type = SCM_RIGHTS, c->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(c), &fd, sizeof(int));
    return sendmsg(sock, &msg, 0) == 1 ? 0 : -1;
}

static int recv_fd(int sock)
{
    char byte, ctl[CMSG_SPACE(sizeof(int))];
    struct iovec io = {&byte, 1};
    struct msghdr msg;
    memset(&msg, 0, sizeof(msg));
    msg.msg_iov = &io, msg.msg_iovlen = 1, msg.msg_control = ctl, msg.msg_controllen = sizeof(ctl);
    if (recvmsg(sock, &msg, 0) != 1) return -1;
    struct cmsghdr *c = CMSG_FIRSTHDR(&msg);
    if (!c || c->cmsg_type != SCM_RIGHTS) return -1;
    int fd;
    memcpy(&fd, CMSG_DATA(c), sizeof(int));
    return fd;
}

static const size_t kBytes = 8u << 20;
static const uint32_t kWords = kBytes / 4;

static int producer(int sock)
{
    uint32_t *buf = nullptr;
    TRY(hipMalloc(reinterpret_cast<void **>(&buf), kBytes));
    hipLaunchKernelGGL(fill, dim3(kWords / 256), dim3(256), 0, nullptr, buf, kWords, 7u);
    TRY(hipDeviceSynchronize());
    int fd = -1;
    TRY(hipMemGetHandleForAddressRange(&fd, reinterpret_cast<hipDeviceptr_t>(buf), kBytes,
                                       hipMemRangeHandleTypeDmaBufFd, 0));
    printf("producer: dma-buf descriptor %d for %zu bytes\n", fd, kBytes);
    if (send_fd(sock, fd) != 0) return 1;
    char ack = 0;
    if (read(sock, &ack, 1) != 1) return 1;          // the consumer has read frame 1
    hipLaunchKernelGGL(fill, dim3(kWords / 256), dim3(256), 0, nullptr, buf, kWords, 99u);
    TRY(hipDeviceSynchronize());                     // frame 2 in the same memory
    if (write(sock, "2", 1) != 1) return 1;
    if (read(sock, &ack, 1) != 1) return 1;
    close(fd);
    TRY(hipFree(buf));
    return 0;
}

static int check(const uint32_t *host, uint32_t seed)
{
    for (uint32_t i = 0; i < kWords; ++i)
        if (host[i] != i * 2654435761u + seed) {
            fprintf(stderr, "consumer: word %u differs (seed %u)\n", i, seed);
            return 1;
        }
    return 0;
}

static int consumer(int sock)
{
    const int fd = recv_fd(sock);
    if (fd < 0) return 1;
    hipExternalMemoryHandleDesc desc;
    memset(&desc, 0, sizeof(desc));
    desc.type = hipExternalMemoryHandleTypeOpaqueFd;
    desc.handle.fd = fd;
    desc.size = kBytes;
    hipExternalMemory_t ext = nullptr;
    TRY(hipImportExternalMemory(&ext, &desc));
    hipExternalMemoryBufferDesc range;
    memset(&range, 0, sizeof(range));
    range.size = kBytes;
    void *mapped = nullptr;
    TRY(hipExternalMemoryGetMappedBuffer(&mapped, ext, &range));
    static uint32_t host[kBytes / 4];
    TRY(hipMemcpy(host, mapped, kBytes, hipMemcpyDeviceToHost));
    if (check(host, 7u)) return 1;
    printf("consumer: frame 1 read through the imported mapping\n");
    if (write(sock, "1", 1) != 1) return 1;
    char go = 0;
    if (read(sock, &go, 1) != 1) return 1;
    TRY(hipMemcpy(host, mapped, kBytes, hipMemcpyDeviceToHost));
    if (check(host, 99u)) return 1;
    printf("consumer: frame 2 read from the same mapping: zero-copy hand-off works\n");
    if (write(sock, "k", 1) != 1) return 1;
    TRY(hipDestroyExternalMemory(ext));
    return 0;
}

int main()
{
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv) != 0) return 2;
    const pid_t child = fork();                      // before any HIP call
    if (child < 0) return 2;
    if (child == 0) {
        close(sv[0]);
        _exit(consumer(sv[1]));
    }
    close(sv[1]);
    const int rc = producer(sv[0]);
    int status = 0;
    waitpid(child, &status, 0);
    printf("producer exit %d, consumer exit %d\n", rc, WIFEXITED(status) ? WEXITSTATUS(status) : -1);
    return rc || !WIFEXITED(status) || WEXITSTATUS(status);
}
