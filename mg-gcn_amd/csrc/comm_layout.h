// comm_layout.h -- host arithmetic of the variable-size exchange, shared by both transports of comm.cpp.
// Header-only and free of HIP / RCCL so that a CPU test can compile and pin it (tests/test_abi_host.py):
// libmggcn_comm.so itself cannot be loaded without a GPU (RCCL initialises on load), and its RCCL branch
// of the all-to-all has not run on more than one GPU yet.
#pragma once

#include <cstddef>

namespace mggcn_layout {

// counts[j * P + k] floats go from rank j to rank k; both sides keep their pieces in rank order:
//   sdis[j * P + k] = where rank j's piece for rank k starts in send[j]
//   rdis[j * P + k] = where the piece FROM rank k starts in recv[j]
inline void alltoallv_displacements(int P, const std::size_t *counts, std::size_t *sdis, std::size_t *rdis) {
    for (int j = 0; j < P; j++) {
        sdis[(std::size_t)j * P] = rdis[(std::size_t)j * P] = 0;
        for (int k = 1; k < P; k++) {
            sdis[(std::size_t)j * P + k] = sdis[(std::size_t)j * P + k - 1] + counts[(std::size_t)j * P + k - 1];
            rdis[(std::size_t)j * P + k] = rdis[(std::size_t)j * P + k - 1] + counts[(std::size_t)(k - 1) * P + j];
        }
    }
}

}  // namespace mggcn_layout
