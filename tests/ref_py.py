"""A SECOND, independently written restatement of the reference's edge-identity path, in plain Python, read straight
from the Go source (not from oracle/sg_oracle.c): it exists only so that tests/test_oracle_vs_python.py can diff the two
on random traces — the reference has no golden vectors for FromUID / ToUID and no Go toolchain is here.

  ClusterInfo tables + processPod / processSvc      aggregator/cluster.go:13-17, aggregator/persist.go:25-72, 81-131
  IntToIPv4 / extractAddressPair                     aggregator/data.go:1751-1767
  setFromToV2 (+ getPodWithIP / getSvcWithIP)        aggregator/data.go:812-870
  ReverseDirection                                   datastore/dto.go:226-231
  Request fields set by processHttpEvent & friends   aggregator/data.go:1208-1249 (StartTime :1219, Latency :1220, HTTPS :1240-1242)
  convertKernelTimeToUserspaceTime                   aggregator/data.go:1740-1743
Reverse DNS (getHostnameFromIP, :1386-1405) is treated as failing, as everywhere in this repository.
"""
from __future__ import annotations

POD, SVC, OUTBOUND = "pod", "service", "outbound"
PROTO_NAMES = {1: "HTTP", 2: "AMQP", 3: "POSTGRES", 4: "HTTP2", 5: "REDIS", 6: "KAFKA", 7: "MYSQL", 8: "MONGO"}
U64 = (1 << 64) - 1


def int_to_ipv4(v: int) -> str:
    # data.go:1751-1758: binary.BigEndian.PutUint32 then net.IP.String()
    return f"{(v >> 24) & 255}.{(v >> 16) & 255}.{(v >> 8) & 255}.{v & 255}"


class Aggregator:
    def __init__(self, first_kernel_ns: int, first_user_ns: int):
        self.pod_ip_to_uid: dict[str, str] = {}
        self.svc_ip_to_uid: dict[str, str] = {}
        self.first_kernel, self.first_user = first_kernel_ns, first_user_ns
        self.rows = []            # what PersistRequest receives: (start_ms, latency, from_type, from_uid, to_type, to_uid, protocol, status, tls)
        self.dropped = 0

    # ---- persist.go ----
    def process_pod(self, event_type: str, uid: str, ip: str):
        if ip == "":                                   # persist.go:37-40
            return
        if event_type in ("ADD", "UPDATE"):            # :55-66
            self.pod_ip_to_uid[ip] = uid
        elif event_type == "DELETE":                   # :67-71
            self.pod_ip_to_uid.pop(ip, None)

    def process_svc(self, event_type: str, uid: str, cluster_ip: str):
        if event_type in ("ADD", "UPDATE"):            # :114-125
            self.svc_ip_to_uid[cluster_ip] = uid
        elif event_type == "DELETE":                   # :126-130
            self.svc_ip_to_uid.pop(cluster_ip, None)

    # ---- data.go:827-870 ----
    def set_from_to_v2(self, saddr: str, daddr: str, host_header: str):
        pod = self.pod_ip_to_uid.get(saddr)
        if pod is None:
            return None                                # "error finding pod with sockets saddr"
        svc = self.svc_ip_to_uid.get(daddr)
        if svc is not None:
            return (POD, pod, SVC, svc)
        pod2 = self.pod_ip_to_uid.get(daddr)
        if pod2 is not None:
            return (POD, pod, POD, pod2)
        if host_header != "":
            return (POD, pod, OUTBOUND, host_header)
        return (POD, pod, OUTBOUND, daddr)             # reverse DNS fails -> the address itself

    # ---- one L7 event of a protocol whose handler keeps it (packed-event view: the payload decisions are the packer's) ----
    def l7(self, saddr: int, daddr: int, host_header: str, status: int, protocol: int, tls: bool, reverse: bool,
           duration_ns: int, write_time_ns: int):
        r = self.set_from_to_v2(int_to_ipv4(saddr), int_to_ipv4(daddr), host_header)
        if r is None:
            self.dropped += 1
            return
        ft, fu, tt, tu = r
        if reverse:                                    # dto.go:226-231 after the join (data.go:1110-1112, 1151-1153)
            ft, fu, tt, tu = tt, tu, ft, fu
        # data.go:1740-1743 with u64 wrap-around, then / 1e6 as int64 (:1219)
        start_ns = (self.first_user - ((self.first_kernel - write_time_ns) & U64)) & U64
        start_ms = start_ns // 1_000_000
        proto = PROTO_NAMES.get(protocol, "")
        if proto == "HTTP" and tls:
            proto = "HTTPS"                            # :1240-1242
        self.rows.append((start_ms, duration_ns, ft, fu, tt, tu, proto, status, tls))


# ---- the per-window edge ledger (round 6: a second definition of what a window's rows are) ------------------------------------------
# The reference ships one row per request (datastore/backend.go:819-847) and keeps no per-edge state; a window's edge rows are the
# builder's aggregation of exactly the requests PersistRequest received in that window: an edge is in a window's
# rows if and only if a request on it arrived in THAT window — whatever the engine keeps from earlier windows (the warm-window state) must
# not show.  "error": HTTP / HTTPS / HTTP2 status >= 500; POSTGRES / REDIS / MYSQL status 2 (ebpf/c/postgres.c:91, redis.c:10, mysql.c:36).
def is_error(proto: str, status: int) -> bool:
    if proto in ("HTTP", "HTTPS", "HTTP2"):
        return status >= 500
    if proto in ("POSTGRES", "REDIS", "MYSQL"):
        return status == 2
    return False


def window_ledger(rows):
    """rows: Aggregator.rows of ONE window -> {(from_type, from_uid, to_type, to_uid): (count, errors, sum_ns, max_ns, sumsq_us)}"""
    d = {}
    for (_start, lat, ft, fu, tt, tu, proto, status, _tls) in rows:
        c, e, s, m, q = d.get((ft, fu, tt, tu), (0, 0, 0, 0, 0))
        us = lat // 1000
        d[(ft, fu, tt, tu)] = (c + 1, e + (1 if is_error(proto, status) else 0), s + lat, max(m, lat), (q + us * us) & U64)
    return d
