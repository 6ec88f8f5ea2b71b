set pagination off
set confirm off
set breakpoint pending on
set height 0
break cdkf_custom_kernel
run
delete 1
break *(&cdkf_custom_kernel + 0x98)
commands
silent
printf "CHECK base(s1:s0)=%#x:%#x total(s23:s22)=%#x:%#x vcc=%#lx exec=%#lx v3:v2(lane0)=%#x:%#x\n", $s1, $s0, $s23, $s22, $vcc, $exec, $v3[0], $v2[0]
continue
end
break *(&cdkf_custom_kernel + 0x9c)
commands
silent
printf "SURPLUS-JUMP taken\n"
continue
end
break *(&cdkf_custom_kernel + 0xb4)
commands
silent
printf "BODY entered\n"
continue
end
continue
quit
