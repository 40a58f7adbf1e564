/* Plain C11 translation unit: proves include/sift_amd.h is a C header and
 * that every declared entry point links from libsift_amd.so.  Built and run
 * by tests/test_abi.py on the CPU; it makes no compute call. */
#include <stdio.h>
#include <string.h>

#include "sift_amd.h"

int main(void) {
    void *fns[] = {
        (void *)sift_strerror, (void *)sift_last_error, (void *)sift_init, (void *)sift_shutdown, (void *)sift_reset_workspace, (void *)sift_release_stream, (void *)sift_profile_end_samples, (void *)sift_profile_end_kernels,
        (void *)sift_crc32_workspace_bytes, (void *)sift_profile_begin, (void *)sift_profile_end,
        (void *)sift_tune, (void *)sift_crc32_compute_device, (void *)sift_crc32_compute_device_sync,
        (void *)sift_crc32_compute_device_static,
        (void *)sift_crc32_compute_host, (void *)sift_digest_new, (void *)sift_digest_update,
        (void *)sift_digest_update_device, (void *)sift_digest_finalize, (void *)sift_crc32_concat,
        (void *)sift_chunk_crc_device, (void *)sift_crc32_batch_device, (void *)sift_read_chunks_device,
        (void *)sift_read_documents_device, (void *)sift_write_chunk_crcs_device, (void *)sift_write_header, (void *)sift_read_header,
        (void *)sift_write_chunk, (void *)sift_read_chunk, (void *)sift_expect_type, (void *)sift_write_document, (void *)sift_read_document,
    };
    unsigned char hdr[SIFT_HEADER_LEN];
    sift_digest d;
    size_t i;
    for (i = 0; i < sizeof fns / sizeof fns[0]; ++i)
        if (!fns[i]) return 1;
    if (sift_write_header(hdr, sizeof hdr) != SIFT_HEADER_LEN) return 2;
    if (sift_read_header(hdr, sizeof hdr) != SIFT_HEADER_LEN) return 3;
    sift_digest_new(&d, 0);
    if (d.value != SIFT_CRC_INIT || sift_digest_finalize(&d) != 0u) return 4; /* crc.rs:41-52 */
    if (sift_crc32_concat(0x80000000u, 0u, 0u) != 0x80000000u) return 5;
    if (strcmp(sift_strerror(SIFT_ERR_READ_CRC_MISMATCH), "CRC mismatch when reading data") != 0)
        return 6; /* container.rs:16 */
    printf("abi_link ok: %u entry points, workspace %zu bytes\n", (unsigned)(sizeof fns / sizeof fns[0]),
           sift_crc32_workspace_bytes());
    return 0;
}
