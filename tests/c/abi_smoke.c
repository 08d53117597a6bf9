/* The C ABI from plain C99: the header must compile without C++, the library must link, and the host-only entry points
 * must work without a GPU (tests/test_abi.py builds and runs this). */
#include <stdio.h>
#include <string.h>
#include "trajnet_hip.h"
#include "trajnet_hip_profile.h"

int main(void) {
    tnp_lstm_model m;
    memset(&m, 0, sizeof(m));
    if (tnp_abi_version() != TNP_ABI_VERSION) { printf("abi version %d != %d\n", tnp_abi_version(), TNP_ABI_VERSION); return 1; }
    if (tnp_abi_sizeof(0) != sizeof(tnp_lstm_model) || tnp_abi_sizeof(1) != sizeof(tnp_lstm_extras) ||
        tnp_abi_sizeof(2) != sizeof(tnp_step_saves) || tnp_abi_sizeof(3) != sizeof(tnp_train_saves) ||
        tnp_abi_sizeof(4) != sizeof(tnp_bwd_sweep) || tnp_abi_sizeof(5) != sizeof(tnp_wgrad_problem) ||
        tnp_abi_sizeof(6) != sizeof(tnp_adam_tensor)) { printf("struct sizes differ between C and the library\n"); return 2; }
    /* an invalid model is rejected with a message, not a crash */
    if (tnp_lstm_workspace_bytes(&m, 16, 2) != 0) { printf("invalid model accepted\n"); return 3; }
    if (tnp_last_error() == NULL || strlen(tnp_last_error()) == 0) { printf("no error message\n"); return 4; }
    if (tnp_wgrad_workspace_bytes(512, 320, 38912) == 0) { printf("wgrad workspace\n"); return 5; }
    printf("ok abi %d, model %zu bytes, sweep %zu bytes\n", tnp_abi_version(), sizeof(tnp_lstm_model), sizeof(tnp_bwd_sweep));
    return 0;
}
