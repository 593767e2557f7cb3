// psim_checkpoint_file.h -- the file side of the checkpoints: header validation and whole
// reads / writes at an offset.  Host-only code without any GPU call, so that it can also be
// built and checked (under the host sanitizers) on a machine without one.
//
// Format: psim_checkpoint_header (64 bytes), then the particles of the whole set as 64-byte
// `Particle` records (reference layout: src/simulation/mod.rs:86-98) in index order.
#ifndef PSIM_CHECKPOINT_FILE_H
#define PSIM_CHECKPOINT_FILE_H

#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "psim.h"

// Provided by the includer: records the message of a failure and returns `code`.
int psim_fail(int code, const char *fmt, ...);

namespace psim_ckpt {

inline constexpr char kCheckpointMagic[8] = {'P', 'S', 'I', 'M', 'C', 'K', 'P', '1'};

struct File {
    int fd = -1;
    ~File()
    {
        if (fd >= 0) (void)close(fd);
    }
};

inline int write_all(int fd, const void *buf, size_t bytes, uint64_t offset, const char *path)
{
    const char *p = static_cast<const char *>(buf);
    while (bytes) {
        const ssize_t w = pwrite(fd, p, bytes, (off_t)offset);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) return psim_fail(PSIM_ERR_IO, "write to %s failed: %s", path, strerror(errno));
        p += w, bytes -= (size_t)w, offset += (uint64_t)w;
    }
    return PSIM_OK;
}

inline int read_all(int fd, void *buf, size_t bytes, uint64_t offset, const char *path)
{
    char *p = static_cast<char *>(buf);
    while (bytes) {
        const ssize_t r = pread(fd, p, bytes, (off_t)offset);
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return psim_fail(PSIM_ERR_IO, "read from %s failed: %s", path, strerror(errno));
        if (r == 0) return psim_fail(PSIM_ERR_IO, "%s is shorter than its header says", path);
        p += r, bytes -= (size_t)r, offset += (uint64_t)r;
    }
    return PSIM_OK;
}

inline int read_header(int fd, const char *path, psim_checkpoint_header *out)
{
    struct stat st;
    if (fstat(fd, &st) != 0) return psim_fail(PSIM_ERR_IO, "stat of %s failed: %s", path, strerror(errno));
    if ((uint64_t)st.st_size < sizeof(*out))
        return psim_fail(PSIM_ERR_IO, "%s is too short to be a checkpoint", path);
    int rc = read_all(fd, out, sizeof(*out), 0, path);
    if (rc != PSIM_OK) return rc;
    if (memcmp(out->magic, kCheckpointMagic, sizeof(kCheckpointMagic)) != 0)
        return psim_fail(PSIM_ERR_IO, "%s is not a checkpoint (or its write was interrupted)", path);
    if (out->version != 1u || out->record_bytes != sizeof(psim_particle))
        return psim_fail(PSIM_ERR_IO, "%s: unsupported checkpoint version %u / record size %u", path,
                    out->version, out->record_bytes);
    const uint64_t room = ((uint64_t)st.st_size - sizeof(*out)) / sizeof(psim_particle);
    if (out->particle_count > room)
        return psim_fail(PSIM_ERR_IO, "%s holds %llu records, its header says %llu", path,
                    (unsigned long long)room, (unsigned long long)out->particle_count);
    return PSIM_OK;
}

inline uint64_t record_offset(uint64_t particle)
{
    return sizeof(psim_checkpoint_header) + particle * sizeof(psim_particle);
}

// The header of a complete file of `total` particles.
inline psim_checkpoint_header make_header(uint64_t total, uint64_t step, uint32_t generation_mode,
                                          uint32_t color_semantics)
{
    psim_checkpoint_header hdr;
    memset(&hdr, 0, sizeof(hdr));
    memcpy(hdr.magic, kCheckpointMagic, sizeof(kCheckpointMagic));
    hdr.version = 1;
    hdr.record_bytes = sizeof(psim_particle);
    hdr.particle_count = total;
    hdr.step = step;
    hdr.generation_mode = generation_mode;
    hdr.color_semantics = color_semantics;
    return hdr;
}

// The begin / commit pair of a file written by several shards (possibly several processes):
// begin leaves no valid header while records change; commit writes the header once every shard
// has reported, after checking that the file really has room for `total` records.  A run that
// is interrupted between the two leaves a file every reader rejects.
inline int begin_file(const char *path)
{
    File f;
    f.fd = open(path, O_WRONLY | O_CREAT | O_CLOEXEC, 0644);
    if (f.fd < 0) return psim_fail(PSIM_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    psim_checkpoint_header zero;
    memset(&zero, 0, sizeof(zero));
    return write_all(f.fd, &zero, sizeof(zero), 0, path);
}

inline int commit_file(const char *path, const psim_checkpoint_header &hdr)
{
    File f;
    f.fd = open(path, O_WRONLY | O_CLOEXEC);
    if (f.fd < 0) return psim_fail(PSIM_ERR_IO, "cannot open %s for writing: %s", path, strerror(errno));
    struct stat st;
    if (fstat(f.fd, &st) != 0) return psim_fail(PSIM_ERR_IO, "stat of %s failed: %s", path, strerror(errno));
    if ((uint64_t)st.st_size != record_offset(hdr.particle_count))
        return psim_fail(PSIM_ERR_IO, "%s is %llu bytes, a set of %llu particles needs %llu: a shard has not written",
                         path, (unsigned long long)st.st_size, (unsigned long long)hdr.particle_count,
                         (unsigned long long)record_offset(hdr.particle_count));
    const int rc = write_all(f.fd, &hdr, sizeof(hdr), 0, path);
    if (rc != PSIM_OK) return rc;
    const int fd = f.fd;
    f.fd = -1;
    if (close(fd) != 0) return psim_fail(PSIM_ERR_IO, "closing %s failed: %s", path, strerror(errno));
    return PSIM_OK;
}

}  // namespace psim_ckpt

#endif  // PSIM_CHECKPOINT_FILE_H
