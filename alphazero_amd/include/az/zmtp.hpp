// zmtp.hpp -- a minimal ZMTP 3.0 PUSH socket (NULL security mechanism), enough
// to be the producer end of the reference's generator->trainer stream:
// ReplayBuffer(GEN) binds a ZMQ_PUSH on tcp://*:5555 and sends 3-frame
// multipart messages (replay.hpp:41-54,85-111); the reference trainer's
// ZMQ_PULL connects to it (train.cpp:56).  libzmq/cppzmq are not assumed to be
// installed where the generator runs, so the wire protocol is written out here
// (ZeroMQ RFC 23/ZMTP 3.0: 64-byte greeting, READY command, short/long frames
// with the MORE flag).
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace az {

class ZmtpPush {
public:
    // endpoint: "tcp://*:5555" or "tcp://127.0.0.1:5555"
    explicit ZmtpPush(const std::string& endpoint) {
        const auto colon = endpoint.rfind(':');
        if (endpoint.compare(0, 6, "tcp://") != 0 || colon == std::string::npos)
            throw std::runtime_error("ZmtpPush: endpoint must be tcp://host:port");
        const std::string host = endpoint.substr(6, colon - 6);
        const int port = std::stoi(endpoint.substr(colon + 1));
        lfd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (lfd_ < 0) throw std::runtime_error("ZmtpPush: socket() failed");
        int one = 1;
        setsockopt(lfd_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in sa{};
        sa.sin_family = AF_INET;
        sa.sin_port = htons((uint16_t)port);
        sa.sin_addr.s_addr = host == "*" ? htonl(INADDR_ANY) : inet_addr(host.c_str());
        if (::bind(lfd_, (sockaddr*)&sa, sizeof sa) != 0 || ::listen(lfd_, 16) != 0) {
            ::close(lfd_);
            throw std::runtime_error("ZmtpPush: cannot bind " + endpoint + ": " + strerror(errno));
        }
        socklen_t sl = sizeof sa;
        getsockname(lfd_, (sockaddr*)&sa, &sl);
        port_ = ntohs(sa.sin_port);
        acceptor_ = std::thread([this] { accept_loop(); });
    }
    ~ZmtpPush() {
        alive_ = false;
        ::shutdown(lfd_, SHUT_RDWR);
        ::close(lfd_);
        if (acceptor_.joinable()) acceptor_.join();
        for (int fd : peers_) ::close(fd);
    }
    int port() const { return port_; }
    size_t peers() {
        std::lock_guard<std::mutex> lk(m_);
        return peers_.size();
    }
    // Send one multipart message (all frames to the same peer, round-robin over
    // peers like ZMQ_PUSH).  Blocks until a peer is connected.
    void send_multipart(const std::vector<std::pair<const void*, size_t>>& frames) {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_.wait(lk, [this] { return !peers_.empty() || !alive_; });
            if (!alive_) throw std::runtime_error("ZmtpPush: closed");
            const size_t k = rr_++ % peers_.size();
            const int fd = peers_[k];
            bool ok = true;
            for (size_t i = 0; i < frames.size() && ok; ++i)
                ok = send_frame(fd, frames[i].first, frames[i].second, i + 1 < frames.size(), false);
            if (ok) return;
            ::close(fd);                       // peer went away mid-message: drop it, resend to another
            peers_.erase(peers_.begin() + (long)k);
        }
    }

private:
    static bool write_all(int fd, const void* p, size_t n) {
        const char* c = (const char*)p;
        while (n) {
            const ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL);
            if (w <= 0) { if (errno == EINTR) continue; return false; }
            c += w; n -= (size_t)w;
        }
        return true;
    }
    static bool read_all(int fd, void* p, size_t n) {
        char* c = (char*)p;
        while (n) {
            const ssize_t r = ::recv(fd, c, n, 0);
            if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
            c += r; n -= (size_t)r;
        }
        return true;
    }
    static bool send_frame(int fd, const void* body, size_t n, bool more, bool command) {
        unsigned char hdr[9];
        size_t hl;
        unsigned char flags = (more ? 0x01 : 0x00) | (command ? 0x04 : 0x00);
        if (n <= 255) { hdr[0] = flags; hdr[1] = (unsigned char)n; hl = 2; }
        else {
            hdr[0] = flags | 0x02;
            for (int i = 0; i < 8; ++i) hdr[1 + i] = (unsigned char)((uint64_t)n >> (56 - 8 * i));
            hl = 9;
        }
        return write_all(fd, hdr, hl) && (n == 0 || write_all(fd, body, n));
    }
    static bool handshake(int fd) {
        unsigned char g[64];
        memset(g, 0, sizeof g);
        g[0] = 0xFF; g[8] = 0x01; g[9] = 0x7F;      // signature
        g[10] = 3; g[11] = 0;                       // ZMTP 3.0
        memcpy(g + 12, "NULL", 4);                  // mechanism, zero padded to 20 bytes
        g[32] = 0;                                  // as-server
        if (!write_all(fd, g, 64)) return false;
        unsigned char pg[64];
        if (!read_all(fd, pg, 64)) return false;
        if (pg[0] != 0xFF || pg[9] != 0x7F || pg[10] < 3) return false;
        // READY command: name, then the Socket-Type property
        std::string body;
        body += (char)5; body += "READY";
        body += (char)11; body += "Socket-Type";
        const unsigned char len4[4] = {0, 0, 0, 4};
        body.append((const char*)len4, 4); body += "PUSH";
        if (!send_frame(fd, body.data(), body.size(), false, true)) return false;
        // the peer's READY (or ERROR) command
        unsigned char fh[2];
        if (!read_all(fd, fh, 2)) return false;
        uint64_t n = fh[1];
        if (fh[0] & 0x02) {
            unsigned char rest[7];
            if (!read_all(fd, rest, 7)) return false;
            n = fh[1];
            for (int i = 0; i < 7; ++i) n = (n << 8) | rest[i];
        }
        if (!(fh[0] & 0x04) || n > 4096) return false;
        std::vector<char> cmd(n);
        if (n && !read_all(fd, cmd.data(), n)) return false;
        return n >= 6 && cmd[0] == 5 && memcmp(cmd.data() + 1, "READY", 5) == 0;
    }
    void accept_loop() {
        while (alive_) {
            pollfd p{lfd_, POLLIN, 0};
            if (::poll(&p, 1, 200) <= 0) continue;
            const int fd = ::accept(lfd_, nullptr, nullptr);
            if (fd < 0) continue;
            int one = 1;
            setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
            if (!handshake(fd)) { ::close(fd); continue; }
            {
                std::lock_guard<std::mutex> lk(m_);
                peers_.push_back(fd);
            }
            cv_.notify_all();
        }
        cv_.notify_all();
    }

    int lfd_ = -1, port_ = 0;
    std::atomic<bool> alive_{true};
    std::thread acceptor_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<int> peers_;
    size_t rr_ = 0;
};

}  // namespace az
