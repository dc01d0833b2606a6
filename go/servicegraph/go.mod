// Stub: this package lives inside the Alaz module tree (github.com/ddosify/alaz/servicegraph) when it is vendored into the
// reference; the replace directive below is what a stand-alone checkout next to an Alaz checkout needs.
module github.com/ddosify/alaz/servicegraph

go 1.22

require github.com/ddosify/alaz v0.0.0

replace github.com/ddosify/alaz => ../../../alaz
