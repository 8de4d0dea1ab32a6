// Rendezvous / coordination key-value store over TCP (component C19/C20 replacement).
//
// The reference starts an in-process gRPC tf.train.Server per task and a parameter-server task that
// blocks in server.join() forever (demo2/train.py:13-24).  Here parameters are replicated on every
// GPU, so the "ps" role shrinks to hosting this tiny store: workers register, exchange bootstrap
// blobs (e.g. the RCCL/torch rendezvous address), publish heartbeats and a stop flag, and barrier.
//
// Wire protocol (little-endian): request  = u8 op | u32 klen | key | u64 vlen | value
//                                response = i64 status/len | payload
// ops: 1 SET, 2 GET (blocking until present or timeout, value = i64 timeout ms), 3 ADD (value = i64
// delta, returns new value), 4 CHECK (non-blocking presence), 5 DELETE, 6 NUMKEYS.
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "dtf_host.h"

namespace {
bool read_all(int fd, void* buf, size_t n) {
  char* p = static_cast<char*>(buf);
  while (n) {
    const ssize_t r = ::recv(fd, p, n, 0);
    if (r <= 0) return false;
    p += r;
    n -= (size_t)r;
  }
  return true;
}
bool write_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n) {
    const ssize_t r = ::send(fd, p, n, MSG_NOSIGNAL);
    if (r <= 0) return false;
    p += r;
    n -= (size_t)r;
  }
  return true;
}

struct Server {
  int listen_fd = -1;
  int port = 0;
  std::atomic<bool> stop{false};
  std::thread accept_th;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::string, std::string> kv;
  std::vector<std::thread> workers;
  std::vector<int> fds;

  void serve(int fd) {
    for (;;) {
      uint8_t op;
      uint32_t klen;
      uint64_t vlen;
      if (!read_all(fd, &op, 1) || !read_all(fd, &klen, 4) || klen > (1u << 20)) break;
      std::string key(klen, '\0');
      if (klen && !read_all(fd, &key[0], klen)) break;
      if (!read_all(fd, &vlen, 8) || vlen > (1ull << 30)) break;
      std::string val(vlen, '\0');
      if (vlen && !read_all(fd, &val[0], vlen)) break;
      int64_t status = 0;
      std::string payload;
      if (op == 1) {
        std::lock_guard<std::mutex> l(mu);
        kv[key] = val;
        cv.notify_all();
      } else if (op == 2) {
        int64_t timeout_ms = -1;
        if (val.size() == 8) memcpy(&timeout_ms, val.data(), 8);
        std::unique_lock<std::mutex> l(mu);
        auto pred = [&] { return stop.load() || kv.count(key) > 0; };
        bool ok = true;
        if (timeout_ms < 0) cv.wait(l, pred);
        else ok = dtf::cv_wait_ms(cv, l, timeout_ms, pred);  // steady clock (system clock only under TSan, dtf_host.h)
        if (ok && kv.count(key)) { payload = kv[key]; status = (int64_t)payload.size(); }
        else status = -1;
      } else if (op == 3) {
        int64_t delta = 0;
        if (val.size() == 8) memcpy(&delta, val.data(), 8);
        std::lock_guard<std::mutex> l(mu);
        int64_t cur = 0;
        auto it = kv.find(key);
        if (it != kv.end() && it->second.size() == 8) memcpy(&cur, it->second.data(), 8);
        cur += delta;
        kv[key] = std::string(reinterpret_cast<const char*>(&cur), 8);
        status = cur;
        cv.notify_all();
      } else if (op == 4) {
        std::lock_guard<std::mutex> l(mu);
        status = kv.count(key) ? 1 : 0;
      } else if (op == 5) {
        std::lock_guard<std::mutex> l(mu);
        status = (int64_t)kv.erase(key);
      } else if (op == 6) {
        std::lock_guard<std::mutex> l(mu);
        status = (int64_t)kv.size();
      } else {
        status = -2;
      }
      if (!write_all(fd, &status, 8)) break;
      if (!payload.empty() && !write_all(fd, payload.data(), payload.size())) break;
    }
    {
      // forget the descriptor before closing it: the number may be reused by the process, and stop() shuts down every
      // descriptor still on the list
      std::lock_guard<std::mutex> l(mu);
      for (size_t i = 0; i < fds.size(); ++i)
        if (fds[i] == fd) { fds.erase(fds.begin() + (long)i); break; }
    }
    ::close(fd);
  }

  void accept_loop() {
    while (!stop.load()) {
      pollfd pfd{listen_fd, POLLIN, 0};
      const int r = ::poll(&pfd, 1, 200);
      if (r <= 0) continue;
      const int fd = ::accept(listen_fd, nullptr, nullptr);
      if (fd < 0) continue;
      int one = 1;
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      std::lock_guard<std::mutex> l(mu);
      fds.push_back(fd);
      workers.emplace_back([this, fd] { serve(fd); });
    }
  }
};

struct Client {
  int fd = -1;
  std::mutex mu;
  int64_t request(uint8_t op, const std::string& key, const void* val, uint64_t vlen, std::string* payload) {
    std::lock_guard<std::mutex> l(mu);
    const uint32_t klen = (uint32_t)key.size();
    if (!write_all(fd, &op, 1) || !write_all(fd, &klen, 4) || !write_all(fd, key.data(), klen) ||
        !write_all(fd, &vlen, 8) || (vlen && !write_all(fd, val, vlen)))
      return -100;
    int64_t status;
    if (!read_all(fd, &status, 8)) return -100;
    if (op == 2 && status > 0 && payload) {
      payload->resize((size_t)status);
      if (!read_all(fd, &(*payload)[0], (size_t)status)) return -100;
    }
    return status;
  }
};
}  // namespace

DTF_HOST_API void* dtf_store_server_start(int port) {
  Server* s = new Server();
  s->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
  if (s->listen_fd < 0) { delete s; return nullptr; }
  int one = 1;
  setsockopt(s->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_addr.s_addr = htonl(INADDR_ANY);
  addr.sin_port = htons((uint16_t)port);
  if (::bind(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0 || ::listen(s->listen_fd, 128) != 0) {
    ::close(s->listen_fd);
    delete s;
    return nullptr;
  }
  socklen_t len = sizeof(addr);
  getsockname(s->listen_fd, reinterpret_cast<sockaddr*>(&addr), &len);
  s->port = ntohs(addr.sin_port);
  s->accept_th = std::thread([s] { s->accept_loop(); });
  return s;
}

DTF_HOST_API int dtf_store_server_port(void* h) { return static_cast<Server*>(h)->port; }

DTF_HOST_API void dtf_store_server_stop(void* h) {
  Server* s = static_cast<Server*>(h);
  s->stop.store(true);
  {
    std::lock_guard<std::mutex> l(s->mu);
    s->cv.notify_all();
    for (int fd : s->fds) ::shutdown(fd, SHUT_RDWR);
  }
  if (s->accept_th.joinable()) s->accept_th.join();
  for (auto& t : s->workers)
    if (t.joinable()) t.join();
  ::close(s->listen_fd);
  delete s;
}

DTF_HOST_API void* dtf_store_connect(const char* host, int port, int timeout_ms) {
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char portstr[16];
  snprintf(portstr, sizeof(portstr), "%d", port);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  for (;;) {
    if (getaddrinfo(host, portstr, &hints, &res) == 0 && res) {
      const int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
      if (fd >= 0) {
        if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
          int one = 1;
          setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
          freeaddrinfo(res);
          Client* c = new Client();
          c->fd = fd;
          return c;
        }
        ::close(fd);
      }
      freeaddrinfo(res);
      res = nullptr;
    }
    if (std::chrono::steady_clock::now() >= deadline) return nullptr;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
  }
}

DTF_HOST_API int64_t dtf_store_set(void* h, const char* key, const void* val, uint64_t n) {
  return static_cast<Client*>(h)->request(1, key, val, n, nullptr);
}

// returns the value length (copied up to cap bytes), -1 on timeout, -100 on connection loss
DTF_HOST_API int64_t dtf_store_get(void* h, const char* key, void* buf, uint64_t cap, int64_t timeout_ms) {
  std::string payload;
  const int64_t st = static_cast<Client*>(h)->request(2, key, &timeout_ms, 8, &payload);
  if (st > 0) memcpy(buf, payload.data(), payload.size() < cap ? payload.size() : (size_t)cap);
  return st;
}

DTF_HOST_API int64_t dtf_store_add(void* h, const char* key, int64_t delta) {
  return static_cast<Client*>(h)->request(3, key, &delta, 8, nullptr);
}

DTF_HOST_API int64_t dtf_store_check(void* h, const char* key) {
  return static_cast<Client*>(h)->request(4, key, nullptr, 0, nullptr);
}

DTF_HOST_API int64_t dtf_store_delete(void* h, const char* key) {
  return static_cast<Client*>(h)->request(5, key, nullptr, 0, nullptr);
}

DTF_HOST_API void dtf_store_close(void* h) {
  Client* c = static_cast<Client*>(h);
  if (c->fd >= 0) ::close(c->fd);
  delete c;
}
