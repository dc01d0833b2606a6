// datastore.hpp — C++ mirror of the reference's plugin boundary, datastore.DataStore
// (datastore/datastore.go:3-20) and its DTOs (datastore/dto.go).  Same method names, argument
// meaning and error convention (0 == nil error; implementations never throw), so that code and
// tests written against the reference's interface read the same here.
//
// The Go toolchain is not available in this build environment; INTEGRATION.md shows the cgo
// decorator a maintainer would add on the Go side.  This header is the host-side equivalent.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace alaz {
namespace datastore {

// event types of the k8s resource calls — aggregator/persist.go:12-16
constexpr const char* ADD = "ADD";
constexpr const char* UPDATE = "UPDATE";
constexpr const char* DELETE_ = "DELETE";

struct Pod {            // dto.go:3-12
    std::string UID, Name, Namespace, Image, IP, OwnerType, OwnerID, OwnerName;
};
struct ServicePort { std::string Name; int32_t Src = 0, Dest = 0; std::string Protocol; };
struct Service {        // dto.go:14-27
    std::string UID, Name, Namespace, Type, ClusterIP;
    std::vector<std::string> ClusterIPs;
    std::vector<ServicePort> Ports;
};
struct ReplicaSet { std::string UID, Name, Namespace, OwnerType, OwnerID, OwnerName; int32_t Replicas = 0; };
struct DaemonSet { std::string UID, Name, Namespace; };
struct StatefulSet { std::string UID, Name, Namespace; };
struct Deployment { std::string UID, Name, Namespace; int32_t Replicas = 0; };
struct AddressIP { std::string Type, ID, Name, Namespace, IP; };
struct AddressPort { int32_t Port = 0; std::string Protocol, Name; };
struct Address { std::vector<AddressIP> IPs; std::vector<AddressPort> Ports; };
struct Endpoints { std::string UID, Name, Namespace; std::vector<Address> Addresses; };
struct ContainerPort { int32_t Port = 0; std::string Protocol; };
struct Container { std::string Name, Namespace, PodUID, Image; std::vector<ContainerPort> Ports; };

struct AliveConnection {   // dto.go:97-106
    int64_t CheckTime = 0;
    std::string FromIP, FromType, FromUID; uint16_t FromPort = 0;
    std::string ToIP, ToType, ToUID; uint16_t ToPort = 0;
};

struct Request {           // dto.go:177-195
    int64_t StartTime = 0;
    uint64_t Latency = 0;  // ns
    std::string FromIP, FromType, FromUID; uint16_t FromPort = 0;
    std::string ToIP, ToType, ToUID; uint16_t ToPort = 0;
    std::string Protocol;
    bool Tls = false, Completed = false;
    uint32_t StatusCode = 0;
    std::string FailReason, Method, Path;
    // dto.go:226-231
    void ReverseDirection() {
        std::swap(FromIP, ToIP); std::swap(FromPort, ToPort); std::swap(FromUID, ToUID); std::swap(FromType, ToType);
    }
};

struct KafkaEvent {        // dto.go:122-142
    int64_t StartTime = 0;
    uint64_t Latency = 0;
    std::string FromIP, FromType, FromUID; uint16_t FromPort = 0;
    std::string ToIP, ToType, ToUID; uint16_t ToPort = 0;
    std::string Topic; uint32_t Partition = 0; std::string Key, Value, Type;
    bool Tls = false;
};

// datastore/datastore.go:3-20
class DataStore {
public:
    virtual ~DataStore() = default;
    virtual int PersistPod(const Pod& pod, const std::string& eventType) = 0;
    virtual int PersistService(const Service& service, const std::string& eventType) = 0;
    virtual int PersistReplicaSet(const ReplicaSet& rs, const std::string& eventType) = 0;
    virtual int PersistDeployment(const Deployment& d, const std::string& eventType) = 0;
    virtual int PersistEndpoints(const Endpoints& e, const std::string& eventType) = 0;
    virtual int PersistContainer(const Container& c, const std::string& eventType) = 0;
    virtual int PersistDaemonSet(const DaemonSet& ds, const std::string& eventType) = 0;
    virtual int PersistStatefulSet(const StatefulSet& ss, const std::string& eventType) = 0;
    virtual int PersistRequest(const Request* request) = 0;
    virtual int PersistKafkaEvent(const KafkaEvent* request) = 0;
    virtual int PersistAliveConnection(const AliveConnection* conn) = 0;
};

// A DataStore that accepts everything and does nothing: stands in for BackendDS where only the
// graph engine is of interest (BackendDS itself — pools, batching, HTTP — is out of scope).
class NullDataStore : public DataStore {
public:
    int PersistPod(const Pod&, const std::string&) override { return 0; }
    int PersistService(const Service&, const std::string&) override { return 0; }
    int PersistReplicaSet(const ReplicaSet&, const std::string&) override { return 0; }
    int PersistDeployment(const Deployment&, const std::string&) override { return 0; }
    int PersistEndpoints(const Endpoints&, const std::string&) override { return 0; }
    int PersistContainer(const Container&, const std::string&) override { return 0; }
    int PersistDaemonSet(const DaemonSet&, const std::string&) override { return 0; }
    int PersistStatefulSet(const StatefulSet&, const std::string&) override { return 0; }
    int PersistRequest(const Request*) override { return 0; }
    int PersistKafkaEvent(const KafkaEvent*) override { return 0; }
    int PersistAliveConnection(const AliveConnection*) override { return 0; }
};

}  // namespace datastore
}  // namespace alaz
