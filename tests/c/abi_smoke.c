/* A plain C consumer of the ABI (no Python, no C++): what a cgo / JNI / ctypes binding sees.
 * Build: gcc -std=c99 -I include tests/c/abi_smoke.c -ldl -o abi_smoke ; run: ./abi_smoke path/to/libdotring_hip.so
 * Exercises only host-side entry points, so it runs without a GPU. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "dotring_hip.h"

typedef const char *(*version_fn)(void);
typedef int (*hash_fn)(int, const uint8_t *, size_t, uint8_t *, size_t);
typedef int (*sqrt_fn)(const uint8_t *, uint8_t *);
typedef int (*compress_fn)(const uint8_t *, int, uint8_t *);
typedef int (*count_fn)(void);

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    version_fn version = (version_fn)dlsym(lib, "dr_version");
    hash_fn hash = (hash_fn)dlsym(lib, "dr_host_hash");
    sqrt_fn fr_sqrt = (sqrt_fn)dlsym(lib, "dr_fr_sqrt");
    compress_fn compress = (compress_fn)dlsym(lib, "dr_g1_compress");
    count_fn device_count = (count_fn)dlsym(lib, "dr_device_count");
    if (!version || !hash || !fr_sqrt || !compress || !device_count) return 4;
    printf("version %s devices %d\n", version(), device_count());
    /* SHA-512("abc") starts with ddaf35a1 */
    uint8_t out[64];
    if (hash(DR_HASH_SHA512, (const uint8_t *)"abc", 3, out, 64) != DR_OK) return 5;
    if (out[0] != 0xdd || out[1] != 0xaf || out[2] != 0x35 || out[3] != 0xa1) return 6;
    /* sqrt(4) = 2 or p - 2 in the Bandersnatch base field */
    uint8_t four[32] = {4}, root[32];
    if (fr_sqrt(four, root) != DR_OK) return 7;
    uint8_t two[32] = {2};
    if (memcmp(root, two, 32) != 0 && root[0] != 0xff) return 8;
    /* a non-residue reports DR_ERR_NOTSQUARE: 5 is the field's non-residue */
    uint8_t five[32] = {5};
    if (fr_sqrt(five, root) != DR_ERR_NOTSQUARE) return 9;
    /* infinity compresses to 0xc0 || 0... */
    uint8_t zero[96] = {0}, c48[48];
    if (compress(zero, 1, c48) != DR_OK || c48[0] != 0xc0) return 10;
    puts("abi ok");
    return 0;
}
