"""TEST INFRASTRUCTURE ONLY. Nothing under vita_amd/ may import this package."""
