// ykh_mesh.hpp -- the TCP mesh between the ranks of one host (ykh_launch.cpp): control plane of the built-in
// test transports (host-staged "tcp", ykh_launch.cpp; device-to-device "ipc", ykh_ipc.cpp) and their scalar all-reduce.
#pragma once
#include <cstddef>
#include <vector>

namespace ykh_mesh {

struct TcpState {
    int rank = 0, nranks = 1;
    std::vector<int> fd;                     // one connected socket per peer (-1 for self)
    std::vector<std::vector<char>> rstage;   // tcp transport: per message receive staging (kept until wait)
    std::vector<void*> rdst;
    std::vector<size_t> rbytes;
};
bool send_all(int fd, const void* p, size_t n);
bool recv_all(int fd, void* p, size_t n);
// full mesh over 127.0.0.1-style single-host addressing; null on failure (ranks on several hosts are refused)
TcpState* tcp_connect_mesh(int rank, int nranks, const char* addr, int base_port);
int tcp_allreduce(void* user, int op, long long* val);       // ykh_allreduce_fn over the mesh; `user` = TcpState*

}  // namespace ykh_mesh
