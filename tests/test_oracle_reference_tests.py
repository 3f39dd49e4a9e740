"""Pin the CPU oracle on the known answers the reference's OWN tests assert, without importing the reference:

  * simulation/commandcontrol_test.py:14-71  — command-and-control walkthrough of ToyCtf, total reward 389.0
  * _env/cyberbattle_env_test.py:41-114      — 57-action Chain-10 script: done at action 56, then
                                               RuntimeError("new episode must be started with env.reset()")
  * simulation/actions_test.py:291-422       — remote / local exploit, connect and prerequisite assertions
The actuator-level calls (cbo_exploit_* / cbo_connect) are AgentActions without the gym env, node and
vulnerability by index, unclamped rewards.
"""
import numpy as np
import pytest

from marlon_amd import flatten as F
from marlon_amd import model as m
from marlon_amd._abi import EnvSpec
from marlon_amd.samples import chainpattern, toy_ctf
from oracle.oracle import Oracle, philox4x32_10

NONE, CREDS, NODES_, ESC, LATERAL, DATA, PROBE_OK, PROBE_FAIL, FAILED = 0, 1, 2, 3, 4, 5, 6, 7, 8


class C2:
    """commandcontrol.CommandControl restated over the oracle's actuator (commandcontrol.py:107-140): total reward
    is the sum of the unclamped ActionResult rewards."""

    def __init__(self, topo):
        self.t = topo
        self.o = Oracle(topo, EnvSpec(n_envs=1, maximum_node_count=topo.n_nodes, maximum_total_credentials=max(8, len(topo.triples))))
        self.total = 0.0

    def n(self, name):
        return self.t.node_ids.index(name)

    def run_attack(self, node, vuln):
        r, k = self.o.exploit_local(self.n(node), self.t.local_vulnerabilities.index(vuln))
        self.total += r
        return k

    def run_remote_attack(self, src, tgt, vuln):
        r, k = self.o.exploit_remote(self.n(src), self.n(tgt), self.t.remote_vulnerabilities.index(vuln))
        self.total += r
        return k

    def connect_and_infect(self, src, tgt, port, cred):
        r, k = self.o.connect(self.n(src), self.n(tgt), self.t.ports.index(port) if port in self.t.ports else 31,
                              self.t.credential_strings.index(cred))
        self.total += r
        return k == LATERAL and r >= 0       # commandcontrol.connect_and_infect returns False unless the move succeeded


def test_toyctf_command_and_control_total_389():
    c = C2(F.flatten(toy_ctf.new_environment()))
    assert c.run_attack("client", "SearchEdgeHistory") == NODES_
    assert c.run_remote_attack("client", "Website", "ScanPageContent") == NODES_
    assert c.run_remote_attack("client", "GitHubProject", "CredScanGitHistory") == CREDS        # needs SasUrlInCommit&GitHub
    assert c.connect_and_infect("client", "AzureStorage", "HTTPS", "SASTOKEN1")
    assert c.run_remote_attack("client", "Website", "ScanPageSource") == NODES_
    assert c.run_remote_attack("client", "Website.Directory", "NavigateWebDirectoryFurther") == CREDS
    assert c.run_remote_attack("client", "Website.Directory", "NavigateWebDirectory") == NODES_
    assert c.run_remote_attack("client", "Sharepoint", "ScanSharepointParentDirectory") == CREDS
    assert c.connect_and_infect("client", "AzureResourceManager", "HTTPS", "ADPrincipalCreds")
    assert c.run_remote_attack("client", "AzureResourceManager", "ListAzureResources") == NODES_
    assert not c.connect_and_infect("client", "AzureVM", "SSH", "ReusedMySqlCred-web")          # blocked by the VM's firewall
    assert c.connect_and_infect("client", "Website", "SSH", "ReusedMySqlCred-web")
    assert c.run_attack("Website", "CredScanBashHistory") == CREDS
    assert not c.connect_and_infect("Website", "Website[user=monitor]", "sudo", "monitorBashCreds")   # port not even declared
    assert not c.connect_and_infect("client", "Website[user=monitor]", "SSH", "monitorBashCreds")     # SSH blocked
    assert c.connect_and_infect("Website", "Website[user=monitor]", "su", "monitorBashCreds")
    assert c.run_attack("Website[user=monitor]", "CredScan-HomeDirectory") == CREDS
    assert c.connect_and_infect("client", "AzureResourceManager", "HTTPS", "azuread_user_credentials") is False  # already owned: REPEAT
    assert c.total == 389.0                                                                       # commandcontrol_test.py:71


CHAIN10_REWARDS = [14, 4, 100, 14, 100, 6, 6, 2, 6, 0, 11, 4, 9, 100, 14, 6, 8, 0, 100, 2, 6, 11, 4, 9, 100, 2, 14, 100, 6, 6, 6, 0, 11, 4,
                   2, 9, 100, 14, 100, 6, 6, 14, 2, 100, 2, 11, 6, 6, 0, 9, 8, 0, 100, 14, 6, 5000]


def test_chain10_script_ends_at_action_56_then_errors(golden_dir):
    z = np.load(f"{golden_dir}/chain10_script.npz")
    o = Oracle(F.flatten(chainpattern.new_environment(10)), EnvSpec(n_envs=1, maximum_node_count=12, maximum_total_credentials=12))
    rewards = []
    for t in range(56):
        out = o.step(z["actions"][t:t + 1])
        rewards.append(out["reward"][0])
        assert out["errors"] == 0 and out["terminated"][0] == (t == 55)
    assert rewards == CHAIN10_REWARDS and sum(rewards) == 6300.0
    out = o.step(np.array([[2, 10, 5, 2, 4]], np.int32))                  # "this is one too many (after done)"
    assert out["errors"] == 1                                              # the reference raises RuntimeError here


def _actions_test_env():
    """The NODES / SAMPLE_VULNERABILITIES / ENV_IDENTIFIERS fixture of actions_test.py:39-212 (data restated);
    'PortWMIOpen' is added to the identifiers so that validation accepts node 'dc'."""
    admin, system = m.AdminEscalation().tag, m.SystemEscalation().tag
    L, R = m.VulnerabilityType.LOCAL, m.VulnerabilityType.REMOTE
    lib = {
        "UACME61": m.VulnerabilityInfo("", L, m.AdminEscalation(), m.Precondition(f"Windows&Win10&(~({admin}|{system}))")),
        "UACME67": m.VulnerabilityInfo("", L, m.SystemEscalation(), m.Precondition(f"Windows&Win10&(~({admin}|{system}))")),
        "MimikatzLogonpasswords": m.VulnerabilityInfo("", L, m.LeakedCredentials([]), m.Precondition(f"Windows&({admin}|{system})")),
        "RDPBF": m.VulnerabilityInfo("", R, m.LateralMove(), m.Precondition("Windows&PortRDPOpen"), cost=1.0),
    }
    win = ["Windows", "Win10", "PortRDPOpen", "PortHTTPOpen", "PortHTTPsOpen"]
    web = [m.ListeningService("RDP"), m.ListeningService("HTTP"), m.ListeningService("HTTPS")]
    A, B = m.RulePermission.ALLOW, m.RulePermission.BLOCK
    nodes = {
        "a": m.NodeInfo(services=web, value=70, properties=list(win), agent_installed=True, vulnerabilities={
            "ListNeighbors": m.VulnerabilityInfo("", L, m.LeakedNodesId(["b", "c", "dc"])),
            "DumpCreds": m.VulnerabilityInfo("", L, m.LeakedCredentials([m.CachedCredential("Sharepoint", "HTTPS", "ADPrincipalCreds"),
                                                                          m.CachedCredential("Sharepoint", "HTTPS", "cred")]))}),
        "b": m.NodeInfo(services=[m.ListeningService("SSH"), m.ListeningService("SQL")], value=80, properties=["Linux", "PortSSHOpen", "PortSQLOpen"]),
        "c": m.NodeInfo(services=web, value=40, properties=list(win), agent_installed=True),
        "dc": m.NodeInfo(services=[m.ListeningService("RDP"), m.ListeningService("WMI")], value=100, properties=["Windows", "Win10", "PortRDPOpen", "PortWMIOpen"]),
        "Sharepoint": m.NodeInfo(
            services=[m.ListeningService("HTTPS", allowedCredentials=["ADPrincipalCreds"])], value=100, properties=["SharepointLeakingPassword"],
            firewall=m.FirewallConfiguration(incoming=[m.FirewallRule("SSH", A), m.FirewallRule("HTTPS", A), m.FirewallRule("HTTP", A), m.FirewallRule("RDP", B)], outgoing=[]),
            vulnerabilities={"ScanSharepointParentDirectory": m.VulnerabilityInfo("", R, m.LeakedCredentials([m.CachedCredential("a", "HTTPS", "ADPrincipalCreds")]), cost=1.0)}),
    }
    ids = m.Identifiers(local_vulnerabilities=["UACME61", "UACME67", "MimikatzLogonpasswords", "ListNeighbors", "DumpCreds"],
                        remote_vulnerabilities=["RDPBF", "ScanSharepointParentDirectory"], ports=["RDP", "HTTP", "HTTPS", "SSH", "SQL", "WMI"],
                        properties=["Linux", "PortSSHOpen", "PortSQLOpen", "Windows", "Win10", "PortRDPOpen", "PortHTTPOpen", "PortHTTPsOpen",
                                    "SharepointLeakingPassword", "PortWMIOpen"])
    return m.Environment(network=m.create_network(nodes), vulnerability_library=lib, identifiers=ids)


@pytest.fixture
def simple():
    t = F.flatten(_actions_test_env())
    return t, Oracle(t, EnvSpec(n_envs=1, maximum_node_count=8, maximum_total_credentials=8))


def test_actions_exploit_local(simple):                      # actions_test.py:327-352
    t, o = simple
    n, lv = t.node_ids.index, t.local_vulnerabilities.index
    r, k = o.exploit_local(n("a"), lv("MimikatzLogonpasswords"))
    assert k == FAILED and r == -20                           # precondition needs an admin/system tag
    r, k = o.exploit_local(n("a"), lv("UACME61"))
    assert k == ESC and o.node_has_tag(n("a"), 2)             # AdminEscalation tag now in node.properties
    r, k = o.exploit_local(n("c"), lv("UACME67"))
    assert k == ESC and o.node_has_tag(n("c"), 3)
    r, k = o.exploit_local(n("a"), lv("MimikatzLogonpasswords"))
    assert k == CREDS                                         # the precondition flipped after the escalation
    r, k = o.exploit_local(n("a"), lv("UACME61"))
    assert k == FAILED                                        # ~(admin|system) is now false


def test_actions_exploit_remote(simple):                     # actions_test.py:291-324
    t, o = simple
    n, lv, rv = t.node_ids.index, t.local_vulnerabilities.index, t.remote_vulnerabilities.index
    o.exploit_local(n("a"), lv("ListNeighbors"))
    r, k = o.exploit_remote(n("a"), n("c"), rv("ScanSharepointParentDirectory"))
    assert k == NONE and r <= 0                               # vulnerability not on that node: outcome None, reward <= 0
    r, k = o.exploit_remote(n("a"), n("dc"), rv("RDPBF"))
    assert k == LATERAL and r == 100 + 7 - 1 and r < 107      # value + first-time bonus - cost
    r, k = o.exploit_remote(n("b"), n("dc"), rv("RDPBF"))
    assert k == NONE and r == -1                              # source not owned (throws_on_invalid_actions=False)


def test_actions_connect(simple):                            # actions_test.py:355-407
    t, o = simple
    n, lv, port, cred = t.node_ids.index, t.local_vulnerabilities.index, t.ports.index, t.credential_strings.index
    o.exploit_local(n("a"), lv("ListNeighbors"))
    r, k = o.connect(n("a"), n("Sharepoint"), port("HTTPS"), cred("ADPrincipalCreds"))
    assert r == -1 and k == NONE                              # target not discovered / credential not gathered yet
    o.exploit_local(n("a"), lv("DumpCreds"))
    r, k = o.connect(n("a"), n("dc"), port("RDP"), cred("cred"))
    assert k == NONE and r <= 0                               # invalid credentials
    r, k = o.connect(n("a"), n("Sharepoint"), port("RDP"), cred("ADPrincipalCreds"))
    assert r < 0                                              # blocking firewall rule
    r, k = o.connect(n("a"), n("Sharepoint"), port("HTTPS"), cred("ADPrincipalCreds"))
    assert r == 100 and k == LATERAL                          # actions_test.py:405


def test_actions_check_prerequisites(simple):                # actions_test.py:410-422
    t, o = simple
    dc = t.node_ids.index("dc")
    assert o.check_prerequisites(dc, t.local_vulnerabilities.index("MimikatzLogonpasswords")) == 0
    assert o.check_prerequisites(dc, t.local_vulnerabilities.index("UACME61")) == 1


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10
    assert philox4x32_10([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
