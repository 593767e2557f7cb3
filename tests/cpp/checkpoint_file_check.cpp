// checkpoint_file_check.cpp -- the file side of the checkpoints (csrc/psim_checkpoint_file.h)
// alone, built with a plain host compiler under AddressSanitizer + UBSan: writes a file in
// pieces and out of order as the shards of a set do, reads it back, and feeds the reader
// every kind of broken file.
//
// usage: checkpoint_file_check <directory>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "psim_checkpoint_file.h"

static char g_message[512];

int psim_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_message, sizeof(g_message), fmt, ap);
    va_end(ap);
    return code;
}

#define REQUIRE(cond)                                                            \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "line %d: %s (%s)\n", __LINE__, #cond, g_message); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

using namespace psim_ckpt;

static int open_rw(const std::string &path) { return open(path.c_str(), O_RDWR | O_CREAT, 0644); }

static int header_status(const std::string &path, psim_checkpoint_header *out)
{
    File f;
    f.fd = open(path.c_str(), O_RDONLY);
    if (f.fd < 0) return -1;
    return read_header(f.fd, path.c_str(), out);
}

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
    const std::string dir = argv[1];
    const uint64_t total = 1000;
    std::vector<psim_particle> records(total);
    for (uint64_t i = 0; i < total; ++i) {
        memset(&records[i], 0, sizeof(psim_particle));
        records[i].position[0] = (float)i;
        records[i].velocity[2] = -(float)i;
        records[i].color[3] = 1.0f;
    }

    // three "shards" write their records out of order, the owner of particle 0 the header
    const std::string path = dir + "/pieces.ckp";
    {
        File f;
        f.fd = open_rw(path);
        REQUIRE(f.fd >= 0);
        const uint64_t cuts[4] = {0, 333, 334, total};
        for (int k = 2; k >= 0; --k)
            REQUIRE(write_all(f.fd, &records[cuts[k]], (cuts[k + 1] - cuts[k]) * sizeof(psim_particle),
                              record_offset(cuts[k]), path.c_str()) == PSIM_OK);
        const psim_checkpoint_header hdr = make_header(total, 77, PSIM_GEN_FILLED, PSIM_COLOR_RESET);
        REQUIRE(write_all(f.fd, &hdr, sizeof(hdr), 0, path.c_str()) == PSIM_OK);
    }
    psim_checkpoint_header hdr;
    REQUIRE(header_status(path, &hdr) == PSIM_OK);
    REQUIRE(hdr.particle_count == total && hdr.step == 77 && hdr.version == 1);
    REQUIRE(hdr.generation_mode == PSIM_GEN_FILLED && hdr.color_semantics == PSIM_COLOR_RESET);
    {
        File f;
        f.fd = open(path.c_str(), O_RDONLY);
        REQUIRE(f.fd >= 0);
        std::vector<psim_particle> back(total);
        REQUIRE(read_all(f.fd, back.data(), total * sizeof(psim_particle), record_offset(0),
                         path.c_str()) == PSIM_OK);
        REQUIRE(memcmp(back.data(), records.data(), total * sizeof(psim_particle)) == 0);
        // a read that runs past the end of the file is reported, not looped on
        REQUIRE(read_all(f.fd, back.data(), 2 * sizeof(psim_particle), record_offset(total - 1),
                         path.c_str()) == PSIM_ERR_IO);
    }

    // begin / commit of a file written by several shards: no valid header from begin until
    // commit, and no commit while the file has not reached the size of the set
    {
        psim_checkpoint_header got;
        REQUIRE(begin_file(path.c_str()) == PSIM_OK);                // over the valid file above
        REQUIRE(header_status(path, &got) == PSIM_ERR_IO);           // records intact, header gone
        REQUIRE(commit_file(path.c_str(), make_header(total, 78, 0, 0)) == PSIM_OK);
        REQUIRE(header_status(path, &got) == PSIM_OK && got.step == 78);
        const std::string part = dir + "/partial.ckp";
        std::remove(part.c_str());
        REQUIRE(begin_file(part.c_str()) == PSIM_OK);
        {
            File f;
            f.fd = open_rw(part);
            REQUIRE(f.fd >= 0);
            REQUIRE(write_all(f.fd, records.data(), 500 * sizeof(psim_particle), record_offset(0),
                              part.c_str()) == PSIM_OK);             // the second shard never wrote
        }
        REQUIRE(commit_file(part.c_str(), make_header(total, 1, 0, 0)) == PSIM_ERR_IO);
        REQUIRE(std::string(g_message).find("has not written") != std::string::npos);
        REQUIRE(header_status(part, &got) == PSIM_ERR_IO);
        REQUIRE(commit_file((dir + "/missing.ckp").c_str(), make_header(0, 0, 0, 0)) == PSIM_ERR_IO);
        REQUIRE(commit_file(path.c_str(), make_header(total, 77, PSIM_GEN_FILLED, PSIM_COLOR_RESET)) == PSIM_OK);
    }

    // broken files
    const std::string bad = dir + "/bad.ckp";
    auto write_bad = [&](const psim_checkpoint_header &h, uint64_t n_records, size_t header_bytes) {
        std::remove(bad.c_str());
        File f;
        f.fd = open_rw(bad);
        if (f.fd < 0) return false;
        if (write_all(f.fd, &h, header_bytes, 0, bad.c_str()) != PSIM_OK) return false;
        return n_records == 0 ||
               write_all(f.fd, records.data(), n_records * sizeof(psim_particle), record_offset(0),
                         bad.c_str()) == PSIM_OK;
    };
    psim_checkpoint_header h = make_header(10, 0, 0, 0);
    REQUIRE(write_bad(h, 0, 8) && header_status(bad, &hdr) == PSIM_ERR_IO);            // too short
    REQUIRE(write_bad(h, 9, sizeof(h)) && header_status(bad, &hdr) == PSIM_ERR_IO);    // a record short
    REQUIRE(write_bad(h, 10, sizeof(h)) && header_status(bad, &hdr) == PSIM_OK);
    h.magic[0] = 0;
    REQUIRE(write_bad(h, 10, sizeof(h)) && header_status(bad, &hdr) == PSIM_ERR_IO);   // zeroed magic
    h = make_header(10, 0, 0, 0);
    h.version = 2;
    REQUIRE(write_bad(h, 10, sizeof(h)) && header_status(bad, &hdr) == PSIM_ERR_IO);
    h = make_header(10, 0, 0, 0);
    h.record_bytes = 48;
    REQUIRE(write_bad(h, 10, sizeof(h)) && header_status(bad, &hdr) == PSIM_ERR_IO);
    h = make_header(~0ull, 0, 0, 0);                                                   // absurd count
    REQUIRE(write_bad(h, 10, sizeof(h)) && header_status(bad, &hdr) == PSIM_ERR_IO);
    REQUIRE(header_status(dir + "/missing.ckp", &hdr) == -1);
    // writing to a descriptor that cannot be written is an error, with the path in the text
    {
        File f;
        f.fd = open(path.c_str(), O_RDONLY);
        REQUIRE(f.fd >= 0);
        REQUIRE(write_all(f.fd, &h, sizeof(h), 0, path.c_str()) == PSIM_ERR_IO);
        REQUIRE(std::string(g_message).find(path) != std::string::npos);
    }
    std::printf("checkpoint file ok\n");
    return 0;
}
