"""Host-side data model for topologies fed to the MI355X step engine.

This is the drop-in mirror of the record types a CyberBattleSim user writes topologies with
(reference: src/CyberBattleSim/cyberbattle/simulation/model.py:63-77 ListeningService,
:104-115 PrivilegeLevel/escalate, :118-198 outcome classes, :208-247 Precondition and
VulnerabilityInfo, :256-305 firewall rules, :308-345 MachineStatus/NodeInfo, :347-362
Identifiers, :377-396 Environment, :403-464 create_network and the identifier inference
helpers).  Same names, same field names, same defaults, so a topology script written for the
reference builds here unchanged; the records are then lowered by marlon_amd.flatten into the
flat table blob the HIP kernels read.  There is no networkx, YAML, plotting or random
labelling here: those are tooling outside the hot path.

Two reference quirks are kept on purpose:
  * ListeningService.sla_weight and NodeInfo.status are class attributes, not constructor
    fields: every service weighs 1.0 and every node starts Running.
  * Rates are carried but never read by the step rules.
"""
from __future__ import annotations

import enum
import random
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Iterable, Iterator, List, NamedTuple, Optional, Tuple, Union

from .precondition import BoolExpr, parse_expression

NodeID = str
ID = str
CredentialID = str
NodeValue = int
PortName = str
VulnerabilityID = str
Probability = float
PropertyName = str

VERSION_TAG = "0.1.0"


@dataclass
class ListeningService:
    name: PortName
    allowedCredentials: List[CredentialID] = field(default_factory=list)
    running: bool = True
    sla_weight = 1.0  # class attribute in the reference too (model.py:75)


class Rates(NamedTuple):
    probingDetectionRate: Probability = 0.0
    exploitDetectionRate: Probability = 0.0
    successRate: Probability = 1.0


class VulnerabilityType(enum.Enum):
    LOCAL = 1
    REMOTE = 2


class PrivilegeLevel(enum.IntEnum):
    NoAccess = 0
    LocalUser = 1
    Admin = 2
    System = 3
    MAXIMUM = 3


def escalate(current_level, escalation_level) -> PrivilegeLevel:
    return PrivilegeLevel(max(int(current_level), int(escalation_level)))


def privilege_tag(level) -> str:
    """Property tag appended to a node when an escalation level is reached (model.py:138-142)."""
    return f"privilege_{int(level)}"


class VulnerabilityOutcome:
    """Base class of what a successful exploit yields."""


class LateralMove(VulnerabilityOutcome):
    success: bool


class CustomerData(VulnerabilityOutcome):
    pass


class PrivilegeEscalation(VulnerabilityOutcome):
    def __init__(self, level: PrivilegeLevel):
        self.level = level

    @property
    def tag(self) -> str:
        return privilege_tag(self.level)


class SystemEscalation(PrivilegeEscalation):
    def __init__(self):
        super().__init__(PrivilegeLevel.System)


class AdminEscalation(PrivilegeEscalation):
    def __init__(self):
        super().__init__(PrivilegeLevel.Admin)


class ProbeSucceeded(VulnerabilityOutcome):
    def __init__(self, discovered_properties: List[PropertyName]):
        self.discovered_properties = discovered_properties


class ProbeFailed(VulnerabilityOutcome):
    pass


class ExploitFailed(VulnerabilityOutcome):
    pass


class CachedCredential(NamedTuple):
    node: NodeID
    port: PortName
    credential: CredentialID


class LeakedCredentials(VulnerabilityOutcome):
    def __init__(self, credentials: List[CachedCredential]):
        self.credentials = credentials


class LeakedNodesId(VulnerabilityOutcome):
    def __init__(self, nodes: List[NodeID]):
        self.nodes = nodes


VulnerabilityOutcomes = Union[LeakedCredentials, LeakedNodesId, PrivilegeEscalation, AdminEscalation,
                              SystemEscalation, CustomerData, LateralMove, ExploitFailed]


class Precondition:
    """Boolean expression over node property names (plus privilege_k tags).

    The reference hands the string to boolean.py 4.0 (model.py:219-223); here it is compiled by
    marlon_amd.precondition, which restates that package's tokenizer and precedence."""

    def __init__(self, expression: Union[BoolExpr, str]):
        if isinstance(expression, BoolExpr):
            self.expression = expression
        else:
            self.expression = parse_expression(str(expression))

    def __repr__(self) -> str:
        return f"Precondition({str(self.expression)!r})"


class VulnerabilityInfo(NamedTuple):
    description: str
    type: VulnerabilityType
    outcome: VulnerabilityOutcome
    precondition: Precondition = Precondition("true")
    rates: Rates = Rates()
    URL: str = ""
    cost: float = 1.0
    reward_string: str = ""


VulnerabilityLibrary = Dict[VulnerabilityID, VulnerabilityInfo]


class RulePermission(enum.Enum):
    ALLOW = 0
    BLOCK = 1


@dataclass(frozen=True)
class FirewallRule:
    port: PortName
    permission: RulePermission
    reason: str = ""


def _stock_rules() -> List[FirewallRule]:
    return [FirewallRule(p, RulePermission.ALLOW) for p in ("RDP", "SSH", "HTTPS", "HTTP")]


@dataclass
class FirewallConfiguration:
    """First rule matching a port wins; a port with no rule is blocked (actions.py:504-515)."""
    outgoing: List[FirewallRule] = field(default_factory=_stock_rules)
    incoming: List[FirewallRule] = field(default_factory=_stock_rules)


class MachineStatus(enum.Enum):
    Stopped = 0
    Running = 1
    Imaging = 2


@dataclass
class NodeInfo:
    services: List[ListeningService]
    vulnerabilities: VulnerabilityLibrary = field(default_factory=dict)
    value: NodeValue = 0
    properties: List[PropertyName] = field(default_factory=list)
    firewall: FirewallConfiguration = field(default_factory=FirewallConfiguration)
    agent_installed: bool = False
    privilege_level: PrivilegeLevel = PrivilegeLevel.NoAccess
    reimagable: bool = True
    last_reimaging: Optional[object] = None
    owned_string: str = ""
    status = MachineStatus.Running  # class attribute in the reference too (model.py:341)
    sla_weight: float = 1.0


class Identifiers(NamedTuple):
    properties: List[PropertyName] = []
    ports: List[PortName] = ["Null"]
    local_vulnerabilities: List[VulnerabilityID] = []
    remote_vulnerabilities: List[VulnerabilityID] = []


class _NodeView:
    """`network.nodes` look-alike: mapping NodeID -> {"data": NodeInfo}, insertion ordered."""

    def __init__(self, store: "OrderedDict[NodeID, dict]"):
        self._store = store

    def __getitem__(self, node_id: NodeID) -> dict:
        return self._store[node_id]

    def __contains__(self, node_id) -> bool:
        return node_id in self._store

    def __iter__(self):
        return iter(self._store)

    def __len__(self) -> int:
        return len(self._store)

    def items(self):
        return self._store.items()

    def keys(self):
        return self._store.keys()


class Network:
    """Ordered node container standing in for the reference's networkx.DiGraph.

    Only the operations the hot path performs on the graph are offered: ordered iteration,
    lookup by id and `has_node`.  Edge annotations (KNOWS / REMOTE_EXPLOIT / LATERAL_MOVE,
    actions.py:96-101,203-222) are rendering data that no rule reads back, so they are not kept.
    """

    def __init__(self):
        self._store: "OrderedDict[NodeID, dict]" = OrderedDict()

    def add_nodes_from(self, pairs: Iterable[Tuple[NodeID, dict]]) -> None:
        for node_id, attrs in pairs:
            self._store.setdefault(node_id, {}).update(attrs)

    def has_node(self, node_id: NodeID) -> bool:
        return node_id in self._store

    @property
    def nodes(self) -> _NodeView:
        return _NodeView(self._store)

    def __len__(self) -> int:
        return len(self._store)


def iterate_network_nodes(network) -> Iterator[Tuple[NodeID, NodeInfo]]:
    for node_id, attrs in network.nodes.items():
        yield node_id, attrs["data"]


@dataclass
class Environment:
    network: object
    vulnerability_library: VulnerabilityLibrary
    identifiers: Identifiers
    version: str = VERSION_TAG

    def nodes(self) -> Iterator[Tuple[NodeID, NodeInfo]]:
        return iterate_network_nodes(self.network)

    def get_node(self, node_id: NodeID) -> NodeInfo:
        return self.network.nodes[node_id]["data"]


def create_network(nodes: Dict[NodeID, NodeInfo]) -> Network:
    net = Network()
    net.add_nodes_from((k, {"data": v}) for k, v in list(nodes.items()))
    return net


# -- identifier inference (model.py:413-464): sorted unions over nodes and the global library --

def collect_ports_from_vuln(vuln: VulnerabilityInfo) -> List[PortName]:
    creds = getattr(vuln.outcome, "credentials", None)
    return [c.port for c in creds] if creds is not None else []


def _as_list(nodes) -> List[Tuple[NodeID, NodeInfo]]:
    return list(nodes)


def collect_vulnerability_ids_from_nodes_bytype(nodes, global_vulnerabilities: VulnerabilityLibrary,
                                                type: VulnerabilityType) -> List[VulnerabilityID]:
    ids = {vid for _, info in _as_list(nodes) for vid, v in info.vulnerabilities.items() if v.type == type}
    ids |= {vid for vid, v in global_vulnerabilities.items() if v.type == type}
    return sorted(ids)


def collect_properties_from_nodes(nodes) -> List[PropertyName]:
    return sorted({p for _, info in _as_list(nodes) for p in info.properties})


def collect_ports_from_nodes(nodes, vulnerability_library: VulnerabilityLibrary) -> List[PortName]:
    nodes = _as_list(nodes)
    ports = {p for v in vulnerability_library.values() for p in collect_ports_from_vuln(v)}
    ports |= {p for _, info in nodes for v in info.vulnerabilities.values() for p in collect_ports_from_vuln(v)}
    ports |= {s.name for _, info in nodes for s in info.services}
    return sorted(ports)


def collect_ports_from_environment(environment: Environment) -> List[PortName]:
    return collect_ports_from_nodes(environment.nodes(), environment.vulnerability_library)


def infer_constants_from_nodes(nodes, vulnerabilities: VulnerabilityLibrary) -> Identifiers:
    nodes = _as_list(nodes)
    return Identifiers(
        properties=collect_properties_from_nodes(nodes),
        ports=collect_ports_from_nodes(nodes, vulnerabilities),
        local_vulnerabilities=collect_vulnerability_ids_from_nodes_bytype(nodes, vulnerabilities, VulnerabilityType.LOCAL),
        remote_vulnerabilities=collect_vulnerability_ids_from_nodes_bytype(nodes, vulnerabilities, VulnerabilityType.REMOTE),
    )


def infer_constants_from_network(network, vulnerabilities: VulnerabilityLibrary) -> Identifiers:
    return infer_constants_from_nodes(iterate_network_nodes(network), vulnerabilities)


# -- random labelling of a graph (model.py:470-539) --

SAMPLE_IDENTIFIERS = Identifiers(
    ports=["RDP", "SSH", "SMB", "HTTP", "HTTPS", "WMI", "SQL"],
    properties=["Windows", "Linux", "HyperV-VM", "Azure-VM", "Win7", "Win10", "PortRDPOpen", "GuestAccountEnabled"])


def assign_random_labels(graph, vulnerabilities: VulnerabilityLibrary = {}, identifiers: Identifiers = SAMPLE_IDENTIFIERS) -> Network:
    """Environment network over the nodes of `graph` (anything with `.nodes` and `.edges()`, e.g. a networkx.DiGraph) with
    randomly drawn properties, firewall rules, a random subset of `vulnerabilities` and, for nodes with out-edges, a
    `RecentlyAccessedMachines` vulnerability leaking their successors.

    Consumes Python's `random` exactly as the reference does (model.py:475-539), so `random.seed(s)` beforehand yields its
    network: the entry node's index; then per node — entry first, the others in graph order — [a value in 0..100, not for
    the entry node], the number of properties and their sample, the library threshold and one draw per library entry, the
    number of outgoing ports and their sample, the same for incoming.  The reference lists a node's successors through a
    Python set of strings, whose order depends on the interpreter's hash seed when there are several; successors are listed in
    edge order here (identical for nodes with at most one successor, and the step rules never depend on that order)."""
    ids = [str(n) for n in graph.nodes]
    edges = [(str(s), str(t)) for (s, t) in graph.edges()]

    def properties() -> List[PropertyName]:
        k = random.randint(0, len(identifiers.properties))
        return list(random.sample(identifiers.properties, k=k))

    def rules() -> List[FirewallRule]:
        k = random.randint(0, len(identifiers.ports))
        return [FirewallRule(port=p, permission=RulePermission.ALLOW) for p in random.sample(identifiers.ports, k=k)]

    def firewall() -> FirewallConfiguration:
        outgoing = rules()
        return FirewallConfiguration(outgoing=outgoing, incoming=rules())

    def library(node_id: NodeID) -> VulnerabilityLibrary:
        threshold = random.random()
        lib = {k: v for (k, v) in vulnerabilities.items() if random.random() > threshold}
        successors = []
        for (s, t) in edges:
            if s == node_id and t not in successors:
                successors.append(t)
        if successors:
            lib["RecentlyAccessedMachines"] = VulnerabilityInfo(description="AzureVM info, including public IP address",
                                                                type=VulnerabilityType.LOCAL, outcome=LeakedNodesId(successors))
        return lib

    entry = ids[random.randrange(len(ids))]
    data: Dict[NodeID, NodeInfo] = {}
    p = properties()
    v = library(entry)
    data[entry] = NodeInfo(services=[], value=0, properties=p, vulnerabilities=v, firewall=firewall(), agent_installed=True,
                           reimagable=False, privilege_level=PrivilegeLevel.Admin)
    for node in ids:
        if node != entry:
            value = random.randint(0, 100)
            p = properties()
            v = library(node)
            data[node] = NodeInfo(services=[], value=value, properties=p, vulnerabilities=v, firewall=firewall(),
                                  agent_installed=False, privilege_level=PrivilegeLevel.NoAccess)
    return create_network({n: data[n] for n in ids})


# -- reading an Environment from the reference's YAML (model.py:545-554, model_test.py:53-106) --

def load_environment_yaml(stream) -> Environment:
    """Read an environment written by the reference (`yaml.dump(env)` after `model.setup_yaml_serializer()`).

    That format is PyYAML's python-object dialect (`!!python/object:cyberbattle.simulation.model.NodeInfo` ...), which the
    reference reads back with the unsafe `yaml.Loader`.  Here the document is read with a SafeLoader that knows exactly
    the tags of the simulation model and of the networkx graph wrapping the nodes and maps them onto THIS module's
    classes: nothing is imported or called by name from the file.  Anchors / aliases are honoured, so rule lists and
    credential lists shared between nodes stay shared objects (the firewall-list aliasing of DESIGN.md).  Edges of the graph
    are dropped like everywhere else in this package; `creationTime` / `lastModified` are ignored."""
    import yaml

    prefix = "cyberbattle.simulation.model."
    plain = {c.__name__: c for c in (ListeningService, FirewallRule, FirewallConfiguration, NodeInfo, LateralMove, CustomerData,
                                     PrivilegeEscalation, SystemEscalation, AdminEscalation, ProbeSucceeded, ProbeFailed,
                                     ExploitFailed, LeakedCredentials, LeakedNodesId)}
    tuples = {c.__name__: c for c in (Rates, CachedCredential, VulnerabilityInfo, Identifiers)}
    enums = {c.__name__: c for c in (RulePermission, PrivilegeLevel, MachineStatus, VulnerabilityType)}

    class Loader(yaml.SafeLoader):
        pass

    def model_class(suffix, table, node):
        if not suffix.startswith(prefix) or suffix[len(prefix):] not in table:
            raise yaml.constructor.ConstructorError(None, None, f"tag {suffix!r} is not part of the environment format", node.start_mark)
        return table[suffix[len(prefix):]]

    def construct_object(loader, suffix, node):
        if suffix == prefix + "Environment":
            state = loader.construct_mapping(node, deep=True)
            return Environment(network=state["network"], vulnerability_library=state.get("vulnerability_library", {}),
                               identifiers=state["identifiers"], version=state.get("version", VERSION_TAG))
        if suffix == "networkx.classes.digraph.DiGraph":
            state = loader.construct_mapping(node, deep=True)
            net = Network()
            net.add_nodes_from((k, {"data": v["data"]}) for k, v in state["_node"].items())
            return net
        cls = model_class(suffix, plain, node)
        obj = cls.__new__(cls)
        state = loader.construct_mapping(node, deep=True) if isinstance(node, yaml.MappingNode) else {}
        if isinstance(state.get("privilege_level"), int):
            state["privilege_level"] = PrivilegeLevel(state["privilege_level"])
        obj.__dict__.update(state)
        return obj

    def construct_new(loader, suffix, node):
        return model_class(suffix, tuples, node)(*loader.construct_sequence(node, deep=True))

    def construct_apply(loader, suffix, node):
        return model_class(suffix, enums, node)(*loader.construct_sequence(node, deep=True))

    Loader.add_multi_constructor("tag:yaml.org,2002:python/object:", construct_object)
    Loader.add_multi_constructor("tag:yaml.org,2002:python/object/new:", construct_new)
    Loader.add_multi_constructor("tag:yaml.org,2002:python/object/apply:", construct_apply)
    Loader.add_constructor("!BooleanExpression", lambda loader, node: Precondition(loader.construct_scalar(node)))
    Loader.add_constructor("!VulnerabilityType", lambda loader, node: VulnerabilityType[loader.construct_scalar(node)])
    env = yaml.load(stream, Loader)
    if not isinstance(env, Environment):
        raise ValueError("the document is not a serialized Environment")
    return env
