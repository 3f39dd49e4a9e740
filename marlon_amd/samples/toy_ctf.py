"""Ten-node capture-the-flag topology.

Same network as the reference table (src/CyberBattleSim/cyberbattle/samples/toyctf/toy_ctf.py:22-191
nodes, :196 inferred identifiers, :199-200 new_environment): node order, values, properties,
services, credentials, firewall rule lists, vulnerability ids/types/costs/outcomes and the one
non-trivial precondition (`SasUrlInCommit&GitHub`, :108) are data that must match for result
parity.  Expressed here as a compact spec table expanded by `_build`; the flattened blob is
pinned against the one flattened from the reference's objects (tests/golden/topology_toyctf.bin).
"""
from __future__ import annotations

from typing import Dict

from .. import model as m

A, B = m.RulePermission.ALLOW, m.RulePermission.BLOCK
STOCK = [("RDP", A), ("SSH", A), ("HTTPS", A), ("HTTP", A)]

# vulnerability spec: id -> (type, outcome spec, precondition or None); every cost is 1.0
#   ("nodes", [ids])  ("creds", [(node, port, credential)])  ("data",)
_SPEC = [
    ("Website", dict(
        value=100, props=["MySql", "Ubuntu", "nginx/1.10.3"],
        services=[("HTTPS", []), ("SSH", ["ReusedMySqlCred-web"])],
        fw_in=STOCK, fw_out=STOCK + [("su", A), ("sudo", A)],
        owned="FLAG: Login using insecure SSH user/password",
        vulns={
            "ScanPageContent": ("R", ("nodes", ["GitHubProject"]), None),
            "ScanPageSource": ("R", ("nodes", ["Website.Directory"]), None),
            "CredScanBashHistory": ("L", ("creds", [("Website[user=monitor]", "SSH", "monitorBashCreds")]), None),
        })),
    ("Website.Directory", dict(
        value=50, props=["Ubuntu", "nginx/1.10.3", "CTFFLAG:Readme.txt-Discover secret data"],
        services=[("HTTPS", [])],
        vulns={
            "NavigateWebDirectoryFurther": ("R", ("creds", [("Website", "MySQL", "ReusedMySqlCred-web")]), None),
            "NavigateWebDirectory": ("R", ("nodes", ["Sharepoint"]), None),
        })),
    ("Website[user=monitor]", dict(
        value=100, props=["MySql", "Ubuntu", "nginx/1.10.3"],
        services=[("SSH", []), ("SSH-key", ["unkownkey"]), ("su", ["monitorBashCreds"])],
        fw_out=STOCK, fw_in=[("SSH", B), ("sudo", B), ("su", A)] + STOCK,
        owned="FLAG User escalation by stealing credentials from bash history",
        vulns={
            "CredScan-HomeDirectory": ("L", ("creds", [("AzureResourceManager[user=monitor]", "HTTPS", "azuread_user_credentials")]), None),
        })),
    ("GitHubProject", dict(
        value=10, props=["GitHub", "SasUrlInCommit"], services=[("GIT", [])],
        vulns={"CredScanGitHistory": ("R", ("creds", [("AzureStorage", "HTTPS", "SASTOKEN1")]), "SasUrlInCommit&GitHub")})),
    ("AzureStorage", dict(
        value=50, props=["CTFFLAG:LeakedCustomerData"], services=[("HTTPS", ["SASTOKEN1"])],
        vulns={"AccessDataWithSASToken": ("R", ("data",), None)})),
    ("Sharepoint", dict(
        value=100, props=["SharepointLeakingPassword"], services=[("HTTPS", [])],
        fw_in=[("SSH", A), ("HTTP", A), ("HTTPS", A)], fw_out=[],
        vulns={"ScanSharepointParentDirectory": ("R", ("creds", [("AzureResourceManager", "HTTPS", "ADPrincipalCreds")]), None)})),
    ("AzureResourceManager", dict(
        value=50, props=["CTFFLAG:LeakedCustomerData2"],
        services=[("HTTPS", ["ADPrincipalCreds", "azuread_user_credentials"])],
        owned="FLAG: Shared credentials with database user - Obtained secrets hidden in Azure Managed Resources",
        vulns={"ListAzureResources": ("R", ("nodes", ["AzureVM"]), None)})),
    ("AzureResourceManager[user=monitor]", dict(
        value=50, props=[], services=[("HTTPS", ["azuread_user_credentials"])],
        owned="More secrets stolen when logged as interactive `monitor` user in Azure with `az`", vulns={})),
    ("AzureVM", dict(
        value=100, props=["CTFFLAG:VMPRIVATEINFO"], services=[("PING", []), ("SSH", [])],
        fw_in=[("SSH", B)], fw_out=[], vulns={})),
    ("client", dict(
        value=0, props=[], services=[], installed=True, reimagable=False,
        vulns={"SearchEdgeHistory": ("L", ("nodes", ["Website"]), None)})),
]


def _rules(spec):
    return [m.FirewallRule(port, perm) for port, perm in spec]


def _outcome(spec) -> m.VulnerabilityOutcome:
    if spec[0] == "nodes":
        return m.LeakedNodesId(list(spec[1]))
    if spec[0] == "creds":
        return m.LeakedCredentials([m.CachedCredential(*c) for c in spec[1]])
    return m.CustomerData()


def _build() -> Dict[m.NodeID, m.NodeInfo]:
    out: Dict[m.NodeID, m.NodeInfo] = {}
    stock = _rules(STOCK)   # ONE list object wherever the spec says exactly STOCK, as the reference's default_allow_rules
                            # (toy_ctf.py:14-19: Website incoming and Website[user=monitor] outgoing are the same list)
    for node_id, s in _SPEC:
        vulns = {}
        for vid, (kind, outcome, pre) in s["vulns"].items():
            kw = dict(description=vid, outcome=_outcome(outcome), cost=1.0,
                      type=m.VulnerabilityType.LOCAL if kind == "L" else m.VulnerabilityType.REMOTE)
            if pre is not None:
                kw["precondition"] = m.Precondition(pre)
            vulns[vid] = m.VulnerabilityInfo(**kw)
        fw = m.FirewallConfiguration()
        if "fw_in" in s:
            fw = m.FirewallConfiguration(incoming=stock if s["fw_in"] is STOCK else _rules(s["fw_in"]),
                                         outgoing=stock if s["fw_out"] is STOCK else _rules(s["fw_out"]))
        out[node_id] = m.NodeInfo(
            services=[m.ListeningService(p, allowedCredentials=list(c)) for p, c in s["services"]],
            vulnerabilities=vulns, value=s["value"], properties=list(s["props"]), firewall=fw,
            agent_installed=s.get("installed", False), reimagable=s.get("reimagable", True),
            owned_string=s.get("owned", ""))
    return out


nodes = _build()
global_vulnerability_library: Dict[m.VulnerabilityID, m.VulnerabilityInfo] = {}
ENV_IDENTIFIERS = m.infer_constants_from_nodes(list(nodes.items()), global_vulnerability_library)


def new_environment() -> m.Environment:
    return m.Environment(network=m.create_network(_build()), vulnerability_library=global_vulnerability_library,
                         identifiers=ENV_IDENTIFIERS)
